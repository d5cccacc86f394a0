// transport.h -- the data plane between GPUs.
//
// Two transports carry array data on an MI355X node (8 GPUs, full xGMI mesh, one link per GPU pair):
//
//   RCCL  (NCCL / NCCL_PL enums, HALO_COMM_NCCL): grouped ncclSend/ncclRecv (or ncclAllToAll when the
//         chunks are uniform and the communicator is the whole world) on the caller's stream;
//         fully stream-asynchronous.  Replaces reference include/internal/comm_routines.h:296-322, 533-584,
//         686-707.
//   PEER  (NVSHMEM* enums; also the MPI_* enums when the library is built without MPI): one-sided copies
//         over xGMI into the receiver's IPC-mapped buffer.  Every pair of GPUs owns a dedicated link, so
//         the P-1 copies of an all-to-all are issued at once (copy engines: each on its own stream / SDMA queue;
//         kernel copies: one launch that feeds every link); the pipelined enums run a staged pipeline with all peers
//         in every stage.  Ordered ON THE STREAM by monotonic flags (call * 16 + step) in a host-pinned board shared
//         by the ranks of the node, or -- opt-in -- in device memory of the poller (sync.hip): no call blocks the
//         host on GPU work, and the sequence can be captured into a hipGraph.  Buffers from cudecompMalloc are pooled
//         with their mappings, and new mappings are verified by page tags.
//         Replaces the reference's NVSHMEM put path (comm_routines.h:122-258) and stands in for
//         CUDA-aware MPI (comm_routines.h:325-413).  Needs the written buffer to be visible to the peer:
//         buffers from cudecompMalloc are mapped once at allocation (NVSHMEM enums require them, as the
//         reference does); the MPI enums take any device buffer and exchange descriptors per call.
//
// A build with MPI=1 adds the ROCm-aware-MPI implementation of the MPI_* enums (transport_mpi.cc).
#pragma once
#include <hip/hip_runtime_api.h>

#include "internal.h"

namespace cudecomp {

std::unique_ptr<Bootstrap> makeWorldBootstrap(MPI_Comm comm, int instance);
MPI_Comm commFromFortran(MPI_Fint f);

// collective: create the RCCL communicator / the peer registry if they will be needed
void prepareTransports(cudecompHandle_t h, bool need_rccl, bool need_peer);

// highest counter value found in row `slot` of the shared board that involves this rank (0 without a board)
uint64_t peerSlotHigh(cudecompHandle_t h, int slot);
// cudecompMalloc calls served from the pool of released workspaces / new IPC mappings found stale (0, 0 without a peer transport)
void peerPoolCounters(cudecompHandle_t h, int64_t* pool_hits, int64_t* stale_mappings, int64_t* pool_bytes = nullptr,
                      int64_t* retired_imports = nullptr);
// Ranks that SHARE a device (test boxes): count the user compute queues the kernel driver holds on this process's GPU
// (sysfs, /sys/class/kfd/kfd/proc/*/queues) and say so once when they exceed the device's hardware queue slots -- the
// driver then time-slices every process of the device.  Best effort, never fails; returns the
// count (-1 if the driver's tables are not readable) and the slots through *slots.
bool peerQueueCensusRequested();  // CUDECOMP_QUEUE_CENSUS=1
int peerQueueCensus(cudecompHandle_t h, bool warn, int* slots = nullptr);
// throws if a device-side wait of an earlier one-sided exchange gave up (dead peer)
void peerCheckStatus(cudecompHandle_t h);
// one-direction copy rate to the next rank through both copy engines (collective; fills h->link_gbps_*)
void peerMeasureLink(cudecompHandle_t h);
// diagnostic: every rank writes a tagged block at several offsets of the NEXT rank's copy of `buffer` (a buffer
// from cudecompMalloc, `bytes` long) and checks what the PREVIOUS rank wrote into its own; returns mismatches
int peerProbe(cudecompHandle_t h, void* buffer, size_t bytes);

void* workspaceAlloc(cudecompHandle_t h, cudecompGridDesc_t gd, size_t bytes);  // collective
// same, without a grid descriptor: peer_capable = map the buffer into the other ranks for one-sided writes
void* workspaceAllocRaw(cudecompHandle_t h, size_t bytes, bool peer_capable);
void workspaceFreeRaw(cudecompHandle_t h, void* ptr);
// collective: really release everything cudecompFree has parked in the workspace pool (cudecompExtTrimWorkspacePool)
void workspaceTrimPool(cudecompHandle_t h);
// mappings of re-created user buffers that are kept open (see PeerContext::map): close all but the newest `keep`
void peerTrimRetiredImports(cudecompHandle_t h, size_t keep);
void workspaceFree(cudecompHandle_t h, cudecompGridDesc_t gd, void* ptr);       // collective

struct ExchangeBuffers {
  char* send;  // base of the send area (device pointer)
  char* recv;  // base of the receive area (device pointer)
};

// State of one stream-ordered one-sided exchange: where this rank's data lands in every member's memory, and the
// device-side call counter the signal / wait kernels compare the board's flags with.
struct PeerCall {
  int nranks = 0;
  std::vector<char*> remote_recv;  // by member: base of its receive area (transposes) / workspace (halos), as mapped here
  std::vector<char*> remote_out;   // by member: base of its output pencil (direct puts)
  bool direct = false;             // agreed by all members: pack straight into the output pencils, no unpack
  unsigned long long* epoch = nullptr;
};
// First step of every one-sided exchange, BEFORE the pack kernels: resolves the peers' buffers, then bumps the call
// counter and publishes "my receive area is free" on `stream`.  rendezvous = false: the buffers must come from
// cudecompMalloc and sit at the same offset everywhere (contract of the reference's NVSHMEM backends; no host
// communication at all).  rendezvous = true: any device buffer; the members exchange buffer descriptors through the
// shared board, which blocks the host until every member has entered the call (the reference's MPI backends block the
// host as well), and agree on `want_direct`.
PeerCall peerBegin(cudecompHandle_t h, cudecompCommInfo& ci, bool rendezvous, const void* recv_area, const void* output,
                   bool want_direct, hipStream_t stream);

// Two-hop relay of a low-fan-out exchange (plan.h RelayPlan): `world` is the descriptor's communicator of all ranks, `call`
// the call state begun on IT (peerBegin with the symmetric workspace).  Grows the handle's relay region when needed
// (collective).  Ordered on `stream`, nothing blocks the host.
bool peerRelayApplies(cudecompHandle_t h, cudecompGridDesc_t gd, const TransposePlan& plan, cudecompTransposeCommBackend_t backend,
                      bool inplace);
bool peerRelayEnsureRegion(cudecompHandle_t h, const RelayPlan& rp, int es);
void peerRelayAlltoall(cudecompHandle_t h, cudecompCommInfo& world, const TransposePlan& plan, const RelayPlan& rp,
                       const ExchangeBuffers& b, int es, const PeerCall& call, hipStream_t stream);

// All-to-all of the plan's chunks among the members of `ci`.  `stream` carries the pack kernels before and
// the unpack kernels after; on return the exchange is ordered on `stream`.  `call` = peerBegin's result for the
// one-sided transport (nullptr for RCCL / MPI).
void alltoallExchange(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                      const ExchangeBuffers& b, int es, cudecompTransposeCommBackend_t backend, const PeerCall* call,
                      hipStream_t stream);

// Fused pack + put (NVSHMEM_SM enum): runs the plan's pack moves with the peers' receive areas -- or, with
// call.direct, their output pencils -- as destinations.
void peerPutExchange(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& plan, void* const bufs[3], int es,
                     const PeerCall& call, hipStream_t stream);

// Staged pipeline of the one-sided transport (pack / send / unpack overlapped stage by stage, all peers in every
// stage).  Runs the plan's pack and unpack moves itself.
bool peerPipelineAvailable(cudecompHandle_t h, const cudecompCommInfo& ci);
void peerStagedExchange(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                        void* const bufs[3], const ExchangeBuffers& b, int es, const PeerCall& call, hipStream_t stream);

// debugging aid (CUDECOMP_DEBUG_VERIFY_EXCHANGE=1): host-synchronous check of what a one-sided exchange delivered
void peerVerifyExchange(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& plan, const ExchangeBuffers& b, int es,
                        bool sender_side_valid, const char* what, hipStream_t stream);

// Per-peer variant used by the pipelined backends: exchange with the given members only.  Waits for
// pack_done[dst] before sending to dst and makes `stream` wait for the arrival of each chunk.
void alltoallExchangePeers(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                           const ExchangeBuffers& b, int es, cudecompTransposeCommBackend_t backend,
                           const std::vector<int>& src_members, const std::vector<int>& dst_members,
                           hipStream_t stream);

// Halo exchange: face i (at send + send_off[i]) goes to neighbour i, slot i (recv + recv_off[i]) is filled by
// neighbour i; neighbour -1 = nothing on that side.  remote_off[i] = offset of MY face inside neighbour i's
// receive buffer (used by the one-sided transport).
struct HaloExchange {
  char* send;
  char* recv;
  i64 send_off[2], recv_off[2], remote_off[2];  // bytes
  i64 bytes;
  int neighbor[2];  // global ranks
  CommAxis comm_axis;  // communicator the neighbours belong to
};
void haloExchange(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x,
                  cudecompHaloCommBackend_t backend, hipStream_t stream);
// Packed halo update with the local phases overlapped with the exchange: the face packs are launched one by one (an
// event each), every face starts travelling as soon as ITS pack is done while the other face is still being packed,
// and (RCCL) a halo slot is unpacked while the other direction is still in flight.  Runs pack, exchange and unpack;
// returns false if this backend has no overlapped variant (the caller then takes the plain path).
bool haloExchangePackedOverlapped(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x, const HaloPlan& plan,
                                  void* const bufs[3], int es, cudecompHaloCommBackend_t backend, hipStream_t stream);

#ifdef CUDECOMP_WITH_MPI
// bootstrap_mpi.cc
bool mpiTransportAvailable(cudecompCommInfo& ci);
void mpiAlltoall(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& plan, const ExchangeBuffers& b, int es,
                 hipStream_t stream);
void mpiHaloExchange(cudecompHandle_t h, const HaloExchange& x, hipStream_t stream);
#endif

}  // namespace cudecomp
