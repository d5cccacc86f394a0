// transpose.cc -- executes transpose and halo plans: bind pointers, launch the move kernels, run the
// exchange.  The algorithmic content lives in plan.cc (what moves where) and kernels.cc / kernels_*.hip (how).
#include <cstdio>

#include "errors.h"
#include "internal.h"
#include "transport.h"
#include "kernels_batch.h"

namespace cudecomp {

namespace {


std::array<int32_t, 3> arr3(const int32_t* p) {
  return p ? std::array<int32_t, 3>{p[0], p[1], p[2]} : std::array<int32_t, 3>{0, 0, 0};
}

// Capture [pack kernel of destination d -> record events[d]] for every destination into an executable graph.
// Returns nullptr (and switches graphs off for this descriptor, with one warning) if the runtime refuses.
hipGraphExec_t capturePackLoop(cudecompHandle_t h, cudecompGridDesc_t gd, const TransposePlan& plan, void* const* bufs,
                               int es) {
  auto give_up = [&](const char* what, hipError_t e) -> hipGraphExec_t {
    (void)hipGetLastError();
    gd->graphs_failed = true;
    if (h->rank == 0)
      fprintf(stderr, "CUDECOMP:WARN: graph capture of the pipelined pack loop failed (%s: %s); continuing without graphs\n",
              what, hipGetErrorString(e));
    return nullptr;
  };
  hipError_t e;
  if (!gd->graph_stream && (e = hipStreamCreateWithFlags(&gd->graph_stream, hipStreamNonBlocking)) != hipSuccess)
    return give_up("hipStreamCreate", e);
  hipStream_t gs = gd->graph_stream;
  if ((e = hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal)) != hipSuccess) return give_up("begin capture", e);
  const char* failed = nullptr;
  try {
    for (const Move3D& m : plan.pack) {
      launchMoves(&m, 1, bufs, es, gs, &h->tuning);
      hipStreamCaptureStatus status;
      unsigned long long id = 0;
      hipGraph_t capturing = nullptr;
      const hipGraphNode_t* deps = nullptr;
      size_t ndeps = 0;
      if ((e = hipStreamGetCaptureInfo_v2(gs, &status, &id, &capturing, &deps, &ndeps)) != hipSuccess) {
        failed = "capture info";
        break;
      }
      hipGraphNode_t record = nullptr;
      if ((e = hipGraphAddEventRecordNode(&record, capturing, deps, ndeps, gd->events[m.peer])) != hipSuccess) {
        failed = "event record node";
        break;
      }
    }
  } catch (const Error&) {
    failed = "kernel launch";
    e = hipErrorUnknown;
  }
  hipGraph_t graph = nullptr;
  const hipError_t end = hipStreamEndCapture(gs, &graph);
  if (failed) {
    if (graph) (void)hipGraphDestroy(graph);
    return give_up(failed, e);
  }
  if (end != hipSuccess) return give_up("end capture", end);
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return give_up("instantiate", e);
  return exec;
}

// which transport carries an exchange that is neither fused nor the pairwise-flag pipeline
ExecPath exchangePath(cudecompHandle_t h, cudecompCommInfo& ci, cudecompTransposeCommBackend_t backend) {
  if (transposeBackendIsRccl(backend)) return PATH_RCCL;
#ifdef CUDECOMP_WITH_MPI
  if (transposeBackendIsMpi(backend) && mpiTransportAvailable(ci)) return PATH_MPI;
#endif
  (void)h;
  (void)ci;
  return PATH_PEER_BARRIER;
}

}  // namespace

namespace {

// the work of one transpose on `stream` (plan already chosen); pev = performance-sample events or nullptr
void executeTranspose(cudecompHandle_t h, cudecompGridDesc_t gd, const TransposePlan& plan,
                      const cudecompGridDesc::TransposeKey& key, void* const bufs[3], int es,
                      cudecompTransposeCommBackend_t backend, bool inplace, bool pipelined, hipEvent_t* pev,
                      hipStream_t stream, bool in_capture = false);

// One-sided backends that need no host communication per call (NVSHMEM / NVSHMEM_PL enums: symmetric workspace, order
// kept by device-side epochs) can run as ONE graph launch: with CUDECOMP_ENABLE_CUDA_GRAPHS=1 the whole operation --
// epoch kernel, packs, per-peer waits / copies / signals on the copy streams, unpacks -- is captured once per (plan,
// buffers) on a private stream and replayed on the caller's stream afterwards (the reference captures only the pack
// loop, src/graph.cc; its exchanges are host calls).  Returns false if the operation was not (or could not be) run
// through a graph.
bool runAsGraph(cudecompHandle_t h, cudecompGridDesc_t gd, const TransposePlan& plan, const cudecompGridDesc::TransposeKey& key,
                void* const bufs[3], int es, cudecompTransposeCommBackend_t backend, bool inplace, bool pipelined,
                hipStream_t stream) {
  if (!h->graphs_enable || gd->graphs_failed || !plan.exchange) return false;
  if (!(backend == CUDECOMP_TRANSPOSE_COMM_NVSHMEM || backend == CUDECOMP_TRANSPOSE_COMM_NVSHMEM_PL)) return false;
  if (peerRelayApplies(h, gd, plan, backend, inplace)) return false;  // (the relay allocates on first use; it runs eagerly)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return false;  // the caller is capturing already: our launches simply join its graph
  }
  // The captured nodes hold the peers' IPC addresses of the workspace: any cudecompMalloc / cudecompFree since the
  // capture may have changed what those addresses mean (a workspace freed and re-created at the same local address is
  // mapped elsewhere by the peers), so the graphs of this descriptor are dropped and captured again.
  if (gd->op_graph_generation != h->region_generation) {
    for (auto& kv : gd->op_graphs) (void)hipGraphExecDestroy(kv.second);
    gd->op_graphs.clear();
    gd->op_graph_seen.clear();
    gd->op_graph_generation = h->region_generation;
  }
  // replays skip peerBegin, and inside a capture its error would read as "capture failed": report a wait kernel of an
  // earlier call that gave up here, as the error it is
  peerCheckStatus(h);
  const cudecompGridDesc::PackGraphKey gkey{key, bufs[0], bufs[1], bufs[2], es};
  auto it = gd->op_graphs.find(gkey);
  bool captured_now = false;
  if (it == gd->op_graphs.end()) {
    auto give_up = [&](const char* what, hipError_t e) {
      (void)hipGetLastError();
      gd->graphs_failed = true;
      if (h->rank == 0)
        fprintf(stderr, "CUDECOMP:WARN: graph capture of a one-sided transpose failed (%s: %s); continuing without graphs\n",
                what, hipGetErrorString(e));
      return false;
    };
    // first use of this (plan, buffers): run it eagerly once -- everything that allocates (device epoch, board
    // registration, copy streams, events) happens here, outside any capture -- and capture it on the next call
    if (!gd->op_graph_seen.insert(gkey).second) {
      hipError_t e;
      if (!gd->graph_stream && (e = hipStreamCreateWithFlags(&gd->graph_stream, hipStreamNonBlocking)) != hipSuccess)
        return give_up("hipStreamCreate", e);
      if ((e = hipStreamBeginCapture(gd->graph_stream, hipStreamCaptureModeThreadLocal)) != hipSuccess)
        return give_up("begin capture", e);
      bool threw = false;
      try {
        executeTranspose(h, gd, plan, key, bufs, es, backend, inplace, pipelined, nullptr, gd->graph_stream, true);
      } catch (const Error&) {
        threw = true;
      }
      hipGraph_t graph = nullptr;
      e = hipStreamEndCapture(gd->graph_stream, &graph);
      if (threw || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return give_up(threw ? "launch inside the capture" : "end capture", threw ? hipErrorUnknown : e);
      }
      hipGraphExec_t exec = nullptr;
      e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) return give_up("instantiate", e);
      it = gd->op_graphs.emplace(gkey, exec).first;
      captured_now = true;  // (the captured body has counted its path already)
    } else {
      return false;
    }
  }
  CD_CHECK_HIP(hipGraphLaunch(it->second, stream));
  gd->graph_launches++;
  if (!captured_now) gd->path_count[pipelined ? PATH_PEER_PIPELINED : PATH_PEER_BARRIER]++;
  return true;
}

}  // namespace

void runTranspose(cudecompHandle_t h, cudecompGridDesc_t gd, TransposeOp op, void* input, void* output, void* work,
                  cudecompDataType_t dtype, const int32_t* in_halo, const int32_t* out_halo, const int32_t* in_pad,
                  const int32_t* out_pad, hipStream_t stream) {
  const int es = elementSize(dtype);
  const bool inplace = (input == output);
  const auto backend = gd->config.transpose_comm_backend;
  TransportTraits traits;
  traits.pipelined = transposeBackendIsPipelined(backend);
  traits.symmetric_recv = usesPeerTransport(h, backend);
  traits.self_exchange = h->self_exchange;

  std::array<int32_t, 12> hp;
  {
    const auto a = arr3(in_halo), b = arr3(out_halo), c = arr3(in_pad), d = arr3(out_pad);
    for (int i = 0; i < 3; ++i) {
      hp[i] = a[i];
      hp[3 + i] = b[i];
      hp[6 + i] = c[i];
      hp[9 + i] = d[i];
    }
  }
  const cudecompGridDesc::TransposeKey key{(int)op, hp, inplace, traits.pipelined, traits.symmetric_recv};
  auto it = gd->transpose_plans.find(key);
  if (it == gd->transpose_plans.end()) {
    const auto& ci = gd->comm((op == OP_X_TO_Y || op == OP_Y_TO_X) ? COMM_COL : COMM_ROW);
    TransposePlan p = buildTransposePlan(gd->shape, h->rank, op, &hp[0], &hp[3], &hp[6], &hp[9], inplace, traits,
                                         ci.npergroup);
    it = gd->transpose_plans.emplace(key, std::move(p)).first;
  }
  const TransposePlan& plan = it->second;
  if (plan.noop) return;

  ensureDevice(h);
  void* bufs[3] = {input, output, work};
  hipEvent_t* pev = perfBeginTranspose(h, gd, (int)op, dtype, hp, inplace, plan.exchange ? plan.pencil_elements_a * es : 0, stream);
  if (runAsGraph(h, gd, plan, key, bufs, es, backend, inplace, traits.pipelined, stream)) {
    // the phases are inside one graph launch: the whole operation counts as exchange time
    perfMark(pev, 1, stream);
    perfMark(pev, 2, stream);
    perfMark(pev, 3, stream);
    return;
  }
  const int64_t direct_before = gd->direct_puts;
  executeTranspose(h, gd, plan, key, bufs, es, backend, inplace, traits.pipelined, pev, stream);
  if (h->debug_verify_exchange && plan.exchange && usesPeerTransport(h, backend)) {
    static const char* names[4] = {"XToY", "YToZ", "ZToY", "YToX"};
    cudecompCommInfo& ci = gd->comm(plan.comm_axis);
    ExchangeBuffers xb;
    xb.send = static_cast<char*>(bufs[plan.send_buf]) + plan.send_base * es;
    xb.recv = static_cast<char*>(bufs[plan.recv_buf]) + plan.recv_base * es;
    // the sender's copy of a chunk is still there afterwards unless the fused put never materialised it or an in-place
    // unpack has overwritten the pencil it was sent from
    const bool fused = backend == CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM;
    const bool sender_valid = !fused && !(inplace && plan.send_buf != BUF_WORK);
    if (gd->direct_puts == direct_before)  // (a direct put has no receive area to look at)
      peerVerifyExchange(h, ci, plan, xb, es, sender_valid, names[(int)op], stream);
  }
}

namespace {

void executeTranspose(cudecompHandle_t h, cudecompGridDesc_t gd, const TransposePlan& plan,
                      const cudecompGridDesc::TransposeKey& key, void* const bufs[3], int es,
                      cudecompTransposeCommBackend_t backend, bool inplace, bool pipelined, hipEvent_t* pev,
                      hipStream_t stream, bool in_capture) {
  void* const input = bufs[0];
  void* const output = bufs[1];
  void* const work = bufs[2];
  if (!plan.exchange) {
    gd->path_count[PATH_LOCAL]++;
    if (plan.rotate && input == output && h->inplace_rotation && rotateSupported(es, plan.rotate_n)) {
      // in place on a cubic 1 x 1 grid: one rotation kernel, one read and one write per element (kernels_rotate.hip)
      launchRotate(input, plan.rotate_n, es, plan.rotate, stream, h->tuning.rotate_walk);
      gd->rotations++;
      perfMark(pev, 1, stream);
      perfMark(pev, 2, stream);
      perfMark(pev, 3, stream);
      return;
    }
    launchMoves(plan.pack.data(), (int)plan.pack.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 1, stream);
    perfMark(pev, 2, stream);
    launchMoves(plan.unpack.data(), (int)plan.unpack.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 3, stream);
    return;
  }

  cudecompCommInfo& ci = gd->comm(plan.comm_axis);
  ExchangeBuffers xb;
  xb.send = static_cast<char*>(bufs[plan.send_buf]) + plan.send_base * es;
  xb.recv = static_cast<char*>(bufs[plan.recv_buf]) + plan.recv_base * es;

  // One-sided transport: the call starts by telling the peers "my receive area is free" -- before the packs, so
  // that their data can start moving while this rank is still packing.  The NVSHMEM / NVSHMEM_PL enums need the
  // workspace from cudecompMalloc at the same offset everywhere (the reference's contract for them) and involve no
  // host communication; the MPI enums and NVSHMEM_SM exchange buffer descriptors per call (any device buffer; SM
  // also agrees on the direct put), which makes the host wait for its peers to ENTER the call, never for GPU work.
  const bool sm = backend == CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM;
  const ExecPath xpath = sm ? PATH_PEER_FUSED : exchangePath(h, ci, backend);
  const bool one_sided = sm || xpath == PATH_PEER_BARRIER;
  PeerCall call;
  // Two-hop relay (opt-in, CUDECOMP_TWO_HOP_RELAY=1): a low-fan-out exchange of the NVSHMEM enum travels through ALL ranks
  // of the node, ordered by the flags of the descriptor's communicator of all ranks (plan.h RelayPlan).
  if (!pipelined && !in_capture && xpath == PATH_PEER_BARRIER && peerRelayApplies(h, gd, plan, backend, inplace)) {
    auto rit = gd->relay_plans.find(key);
    if (rit == gd->relay_plans.end()) {
      TransportTraits traits;
      traits.pipelined = false;
      traits.symmetric_recv = true;
      const auto& hp = std::get<1>(key);
      rit = gd->relay_plans.emplace(key, buildRelayPlan(gd->shape, h->nranks, h->rank, (TransposeOp)std::get<0>(key), &hp[0], &hp[3],
                                                        &hp[6], &hp[9], inplace, traits, ci.npergroup)).first;
    }
    const RelayPlan& rp = rit->second;
    if (rp.applies && peerRelayEnsureRegion(h, rp, es)) {
      // The relay region is ONE per handle and its slots are indexed by (source, chunk) only: relayed transposes of
      // different descriptors or streams must not overlap on a rank.  Every relayed call therefore starts behind the
      // previous one of this handle (an event per call); a peer's scatter into my region waits for my "begun", which I
      // raise behind that event, i.e. after my previous forward has read the slots.
      if (h->relay_last_call) CD_CHECK_HIP(hipStreamWaitEvent(stream, h->relay_last_call, 0));
      else CD_CHECK_HIP(hipEventCreateWithFlags(&h->relay_last_call, hipEventDisableTiming));
      call = peerBegin(h, gd->world, false, xb.recv, output, false, stream);
      gd->path_count[xpath]++;
      gd->relayed++;
      launchMoves(plan.pack.data(), (int)plan.pack.size(), bufs, es, stream, &h->tuning);
      perfMark(pev, 1, stream);
      peerRelayAlltoall(h, gd->world, plan, rp, xb, es, call, stream);
      perfMark(pev, 2, stream);
      launchMoves(plan.unpack.data(), (int)plan.unpack.size(), bufs, es, stream, &h->tuning);
      perfMark(pev, 3, stream);
      CD_CHECK_HIP(hipEventRecord(h->relay_last_call, stream));
      return;
    }
  }
  if (one_sided) {
    const bool rendezvous = sm || !transposeBackendIsPeer(backend);
    const bool want_direct = sm && h->direct_put && !inplace && !plan.direct.empty();
    call = peerBegin(h, ci, rendezvous, xb.recv, output, want_direct, stream);
  }

  if (sm) {
    // compute-unit driven: pack straight into the peers' receive areas, or (direct) into their output pencils
    gd->path_count[PATH_PEER_FUSED]++;
    if (call.direct) gd->direct_puts++;
    perfMark(pev, 1, stream);  // pack and exchange are one fused phase here: all of it counts as exchange
    peerPutExchange(h, ci, plan, bufs, es, call, stream);
    perfMark(pev, 2, stream);
    if (!call.direct) launchMoves(plan.unpack.data(), (int)plan.unpack.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 3, stream);
    return;
  }
  const bool engines_asked_for = h->peer_copy_engine_pinned && h->peer_copy_engine == 0;  // CUDECOMP_PEER_COPY_ENGINE=sdma
  if (!pipelined && xpath == PATH_PEER_BARRIER && transposeBackendIsPeer(backend) && !h->self_exchange && !engines_asked_for &&
      plan.pencil_elements_a * es <= h->fuse_small_bytes) {
    // Small exchanges of the NVSHMEM enum are latency-bound: eight launches and two cross-stream hand-offs (pack, ready wait,
    // copy kernel on the copy stream, signal, self copy, landed wait, unpack) against six launches on ONE stream when the
    // pack kernel stores straight into the peers' receive areas, as NVSHMEM_SM does (16^3 fp64 on 2 / 4 ranks: 66-113 us ->
    // the fused put's 14-19 us, profiles/r04_flags_latency.json).  Same contract (symmetric workspace), same result.
    gd->path_count[xpath]++;
    perfMark(pev, 1, stream);
    peerPutExchange(h, ci, plan, bufs, es, call, stream);
    perfMark(pev, 2, stream);
    launchMoves(plan.unpack.data(), (int)plan.unpack.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 3, stream);
    return;
  }
  if (!pipelined) {
    gd->path_count[xpath]++;
    launchMoves(plan.pack.data(), (int)plan.pack.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 1, stream);
    alltoallExchange(h, gd, ci, plan, xb, es, backend, one_sided ? &call : nullptr, stream);
    perfMark(pev, 2, stream);
    launchMoves(plan.unpack.data(), (int)plan.unpack.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 3, stream);
    return;
  }

  if (one_sided) {
    // staged pipeline: pack / send / unpack overlap stage by stage with all peers in every stage (transport.cc)
    gd->path_count[PATH_PEER_PIPELINED]++;
    perfMark(pev, 1, stream);
    peerStagedExchange(h, gd, ci, plan, bufs, xb, es, call, stream);
    perfMark(pev, 2, stream);
    perfMark(pev, 3, stream);
    return;
  }
  // Per-peer pipeline (RCCL / MPI): pack chunk by chunk (an event per destination), then walk the pairwise schedule:
  // exchange with one peer on the side stream while the previous peer's chunk is being unpacked.
  const int P = plan.nranks;
  if ((int)gd->events.size() < P) {
    const size_t old = gd->events.size();
    gd->events.resize(P);
    for (size_t i = old; i < gd->events.size(); ++i)
      CD_CHECK_HIP(hipEventCreateWithFlags(&gd->events[i], hipEventDisableTiming));
  }
  perfMark(pev, 1, stream);
  if (!plan.pack.empty()) {
    // With graphs enabled the loop is captured once on a private stream -- each destination's kernel followed by an
    // event-record NODE hanging off it, so the side stream can wait on the per-peer events after the launch --
    // and replayed as ONE graph launch on later calls with the same buffers.
    hipGraphExec_t exec = nullptr;
    if (h->graphs_enable && !gd->graphs_failed && plan.pack.size() > 1 && !in_capture) {
      const cudecompGridDesc::PackGraphKey gkey{key, input, output, work, es};
      auto git = gd->pack_graphs.find(gkey);
      if (git != gd->pack_graphs.end()) {
        exec = git->second;
      } else {
        exec = capturePackLoop(h, gd, plan, bufs, es);
        if (exec) gd->pack_graphs.emplace(gkey, exec);
      }
    }
    if (exec) {
      CD_CHECK_HIP(hipGraphLaunch(exec, stream));
      gd->graph_launches++;
    } else {
      for (const Move3D& m : plan.pack) {
        launchMoves(&m, 1, bufs, es, stream, &h->tuning);
        CD_CHECK_HIP(hipEventRecord(gd->events[m.peer], stream));
      }
    }
  } else {
    for (int d = 0; d < P; ++d) CD_CHECK_HIP(hipEventRecord(gd->events[d], stream));
  }
  gd->path_count[xpath]++;
  for (int j = 0; j < P; ++j) {
    const int src = (j == 0) ? plan.comm_rank : plan.schedule_src[j];
    const int dst = (j == 0) ? plan.comm_rank : plan.schedule_dst[j];
    alltoallExchangePeers(h, gd, ci, plan, xb, es, backend, {src}, {dst}, stream);
    for (const Move3D& m : plan.unpack)
      if (m.peer == src) launchMoves(&m, 1, bufs, es, stream, &h->tuning);
  }
  // phases interleave in the per-peer pipeline: report the whole operation as exchange time
  perfMark(pev, 2, stream);
  perfMark(pev, 3, stream);
}

}  // namespace

void runHalo(cudecompHandle_t h, cudecompGridDesc_t gd, int axis, void* input, void* work, cudecompDataType_t dtype,
             const int32_t* halo, const bool* periods, int dim, const int32_t* pad, hipStream_t stream) {
  const int es = elementSize(dtype);
  const auto backend = gd->config.halo_comm_backend;
  const bool force_packed = usesPeerTransport(h, backend);

  const auto hh = arr3(halo), pp = arr3(pad);
  std::array<bool, 3> per{false, false, false};
  if (periods)
    for (int i = 0; i < 3; ++i) per[i] = periods[i];
  const cudecompGridDesc::HaloKey key{axis, dim, {hh[0], hh[1], hh[2], pp[0], pp[1], pp[2]}, per, force_packed};
  auto it = gd->halo_plans.find(key);
  if (it == gd->halo_plans.end()) {
    HaloPlan p = buildHaloPlan(gd->shape, h->rank, axis, dim, hh.data(), per.data(), pp.data(), force_packed, h->self_exchange);
    it = gd->halo_plans.emplace(key, std::move(p)).first;
  }
  const HaloPlan& plan = it->second;
  if (plan.kind == HaloPlan::NONE) return;

  ensureDevice(h);
  void* bufs[3] = {input, input, work};
  int64_t wire_bytes = 0;
  if (plan.kind != HaloPlan::SELF_PERIODIC)
    for (int i = 0; i < 2; ++i)
      if (plan.neighbor[i] >= 0) wire_bytes += plan.face_elements * es;
  hipEvent_t* pev = perfBeginHalo(h, gd, axis, dim, dtype, hh, per, pp, wire_bytes, stream);
  if (plan.kind == HaloPlan::SELF_PERIODIC) {
    launchMoves(plan.pre.data(), (int)plan.pre.size(), bufs, es, stream, &h->tuning);
    perfMark(pev, 1, stream);
    perfMark(pev, 2, stream);
    perfMark(pev, 3, stream);
    return;
  }
  HaloExchange x;
  x.send = x.recv = static_cast<char*>(bufs[plan.xbuf]);
  for (int i = 0; i < 2; ++i) {
    x.send_off[i] = plan.send_off[i] * es;
    x.recv_off[i] = plan.recv_off[i] * es;
    x.remote_off[i] = plan.recv_off[1 - i] * es;  // my low face fills the low neighbour's HIGH halo slot
    x.neighbor[i] = plan.neighbor[i];
  }
  x.bytes = plan.face_elements * es;
  x.comm_axis = plan.comm_axis;
  // packed faces: pack, exchange and unpack overlap face by face (env CUDECOMP_DISABLE_HALO_OVERLAP=1 restores the
  // plain pack -> exchange -> unpack sequence of the reference, halo.h:200-260)
  if (plan.kind == HaloPlan::PACKED && !h->halo_overlap_disable) {
    perfMark(pev, 1, stream);
    if (haloExchangePackedOverlapped(h, gd, x, plan, bufs, es, backend, stream)) {
      perfMark(pev, 2, stream);
      perfMark(pev, 3, stream);
      return;
    }
  }
  if (plan.kind == HaloPlan::PACKED) launchMoves(plan.pre.data(), (int)plan.pre.size(), bufs, es, stream, &h->tuning);
  perfMark(pev, 1, stream);
  haloExchange(h, gd, x, backend, stream);
  perfMark(pev, 2, stream);
  if (plan.kind == HaloPlan::PACKED) launchMoves(plan.post.data(), (int)plan.post.size(), bufs, es, stream, &h->tuning);
  perfMark(pev, 3, stream);
}

}  // namespace cudecomp
