/*
 * cudecomp_ext.h -- entry points of libcudecomp.so that are NOT part of the cuDecomp API.
 *
 * They exist for test harnesses and tools: (1) running one strided block move through the HIP kernel
 * layer, so kernel parity can be tested shape by shape; (2) reading the plan the library would execute
 * for a transpose / halo update as plain data, so the host logic can be checked without a GPU (the
 * multi-process CPU tests execute these plans with numpy + torch.distributed/gloo and compare against
 * the oracle).  Solvers never need this header.
 */
#ifndef CUDECOMP_EXT_H
#define CUDECOMP_EXT_H

#include "cudecomp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CUDECOMP_EXT_MAX_MEMBERS 64

/* dst[dst_off + k0*ds[0] + k1*ds[1] + k2*ds[2]] = src[src_off + k0*ss[0] + k1*ss[1] + k2*ss[2]], in elements.
 * Buffers: 0 = input pencil, 1 = output pencil, 2 = workspace. */
typedef struct {
  int32_t src_buf, dst_buf;
  int64_t src_off, dst_off;
  int64_t extent[3], ss[3], ds[3];
  int32_t peer, row_pitch; /* row_pitch > 0: the move writes whole interior rows of the destination pencil, whose row pitch this is (csrc/plan.h dst_row_pitch) */
} cudecompExtMove_t;

typedef struct {
  int32_t noop;     /* nothing to do */
  int32_t exchange; /* an all-to-all happens between pack and unpack */
  int32_t comm_axis /* 0 column, 1 row */, nranks, comm_rank;
  int32_t send_buf, recv_buf;
  int32_t n_pack, n_unpack;
  int32_t rotate;  /* single-rank in place on a cubic halo-free grid: +1 / -1 = the operation is the in-place rotation new[p0,p1,p2] = old[p2,p0,p1] / its inverse (pack + unpack stay as the staged alternative); 0 = no such form */
  int64_t send_base, recv_base; /* elements */
  int64_t send_cnt[CUDECOMP_EXT_MAX_MEMBERS], send_off[CUDECOMP_EXT_MAX_MEMBERS];
  int64_t recv_cnt[CUDECOMP_EXT_MAX_MEMBERS], recv_off[CUDECOMP_EXT_MAX_MEMBERS];
  int64_t remote_recv_off[CUDECOMP_EXT_MAX_MEMBERS];
  int32_t member_global_rank[CUDECOMP_EXT_MAX_MEMBERS];
  int32_t schedule_dst[CUDECOMP_EXT_MAX_MEMBERS];
  cudecompExtMove_t pack[CUDECOMP_EXT_MAX_MEMBERS];
  cudecompExtMove_t unpack[CUDECOMP_EXT_MAX_MEMBERS];
  /* direct-to-destination put (one-sided transports, out of place): direct[j] moves the slab of MY input (buffer 0) that
   * belongs to member direct[j].peer straight into THAT member's output pencil (dst_buf = 1, offsets and strides in
   * the peer's buffer); no receive area, no unpack.  n_direct = 0: the plan has no such form. */
  int32_t n_direct, reserved2;
  cudecompExtMove_t direct[CUDECOMP_EXT_MAX_MEMBERS];
  /* staged exchange of the one-sided pipelined transports: every chunk is cut into K <= stage_limit ranges along the
   * global axis stage_axis (its slowest wire dim): range k of the chunk for member d is elements
   * [send_n[d]*k/K, send_n[d]*(k+1)/K) * (send_cnt[d]/send_n[d]) of that chunk -- contiguous -- and the same range of
   * extent[stage_axis] of its pack move; likewise recv_n / recv_cnt / unpack on the receiving side */
  int32_t stage_axis, reserved3;
  int64_t stage_limit;
  int64_t send_n[CUDECOMP_EXT_MAX_MEMBERS], recv_n[CUDECOMP_EXT_MAX_MEMBERS];
} cudecompExtTransposePlan_t;

typedef struct {
  int32_t kind; /* 0 none, 1 periodic self copy, 2 packed, 3 direct */
  int32_t comm_axis, neighbor[2];
  int32_t xbuf, n_pre, n_post, reserved;
  int64_t face_elements, send_off[2], recv_off[2];
  cudecompExtMove_t pre[2], post[2];
} cudecompExtHaloPlan_t;

/* op: 0 XToY, 1 YToZ, 2 ZToY, 3 YToX.  The plan is the one cudecompTranspose* would run for the grid
 * descriptor's backend (or for `backend_override` if non-zero). */
cudecompResult_t cudecompExtGetTransposePlan(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, int32_t op,
                                             const int32_t input_halo_extents[], const int32_t output_halo_extents[],
                                             const int32_t input_padding[], const int32_t output_padding[],
                                             bool inplace, int32_t backend_override,
                                             cudecompExtTransposePlan_t* plan);
cudecompResult_t cudecompExtGetHaloPlan(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, int32_t axis,
                                        const int32_t halo_extents[], const bool halo_periods[], int32_t dim,
                                        const int32_t padding[], int32_t backend_override, cudecompExtHaloPlan_t* plan);

/* Stateless planner: the plan rank `rank` of a `pdims[0] x pdims[1]` grid would run, computed without any
 * communicator, handle or device -- one process can therefore build the plans of EVERY rank and execute the whole
 * exchange on host arrays (tests/test_plan_sim.py drives this with randomly drawn decompositions).
 * gdims_dist entries <= 0 mean "same as gdims"; mem_order rows must be permutations.  pipelined / symmetric_recv /
 * npergroup are the transport traits the executor would pass (NCCL: 0/0, *_PL: 1/x, peer transport: x/1). */
typedef struct {
  int32_t gdims[3], gdims_dist[3], pdims[2], col_major;
  int32_t mem_order[3][3]; /* [pencil axis][memory position] -> global axis */
} cudecompExtGridSpec_t;
cudecompResult_t cudecompExtPlanTranspose(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t op,
                                          const int32_t input_halo_extents[], const int32_t output_halo_extents[],
                                          const int32_t input_padding[], const int32_t output_padding[], bool inplace,
                                          int32_t pipelined, int32_t symmetric_recv, int32_t npergroup,
                                          cudecompExtTransposePlan_t* plan);
/* Two-hop relay of a low-fan-out exchange over all ranks of the node (csrc/plan.h RelayPlan; CUDECOMP_TWO_HOP_RELAY=1): the
 * relay moves of `rank` for transpose `op` of the decomposition `grid` -- step 1 "scatter" (from the start of my send area)
 * and step 2 "forward" (from my relay region), each move into the relay region (to_relay = 1, offset in elements) or the
 * receive area (to_relay = 0, offset relative to the receive area) of GLOBAL rank dst_rank.  applies = 0: the exchange
 * of this op is not worth relaying (or is local).  Stateless, like cudecompExtPlanTranspose (symmetric_recv = 1 plan). */
#define CUDECOMP_EXT_MAX_RELAY_MOVES (2 * CUDECOMP_EXT_MAX_MEMBERS * 4)
typedef struct {
  int32_t dst_rank, to_relay;
  int64_t src_off, dst_off, count;
} cudecompExtRelayMove_t;
typedef struct {
  int32_t applies, nranks, slots_per_source, n_scatter, n_forward, reserved;
  int64_t slot_elements, relay_elements;
  cudecompExtRelayMove_t scatter[CUDECOMP_EXT_MAX_RELAY_MOVES], forward[CUDECOMP_EXT_MAX_RELAY_MOVES];
} cudecompExtRelayPlan_t;
cudecompResult_t cudecompExtPlanRelay(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t op, const int32_t in_halo[],
                                      const int32_t out_halo[], const int32_t in_pad[], const int32_t out_pad[], bool inplace,
                                      cudecompExtRelayPlan_t* plan);

cudecompResult_t cudecompExtPlanHalo(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis,
                                     const int32_t halo_extents[], const bool halo_periods[], int32_t dim,
                                     const int32_t padding[], int32_t force_packed, cudecompExtHaloPlan_t* plan);

/* Stateless geometry queries on a grid spec (no handle, no communicator): what cudecompGetPencilInfo,
 * cudecompGetShiftedRank, cudecompGetTransposeWorkspaceSize and cudecompGetHaloWorkspaceSize would answer on `rank`. */
cudecompResult_t cudecompExtPencilInfo(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis,
                                       const int32_t halo_extents[], const int32_t padding[],
                                       cudecompPencilInfo_t* pencil_info);
cudecompResult_t cudecompExtShiftedRank(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis, int32_t dim,
                                        int32_t displacement, bool periodic, int32_t* shifted_rank);
cudecompResult_t cudecompExtWorkspaceSizes(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis,
                                           const int32_t halo_extents[], int64_t* transpose_workspace,
                                           int64_t* halo_workspace);

/* Runs the LOCAL phases (bit 0 of `phases`: pack, bit 1: unpack) of the transpose that rank `rank` of `grid` would
 * run, on the current device and with the executor's own launches, without any exchange: times the kernels at the
 * per-rank shapes of a multi-GPU configuration on one GPU (scripts/probe/local_phases.py).  No halos / padding;
 * input == output plans the in-place form.  Buffers hold that rank's pencils and workspace (es bytes per element). */
cudecompResult_t cudecompExtRunLocalPhases(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t op, int32_t pipelined,
                                           int32_t symmetric_recv, int32_t phases, void* input, void* output, void* work,
                                           int32_t es, hipStream_t stream);

/* Averages over the retained samples (CUDECOMP_PERFORMANCE_REPORT_SAMPLES, all configurations) of one transpose op,
 * recorded when CUDECOMP_ENABLE_PERFORMANCE_REPORT=1 was set at cudecompInit (0 calls otherwise).  Synchronises the device.  exchange_ms is the all-to-all
 * (including the device-side waits for the peers' flags); per-peer pipelined backends report the
 * whole operation as exchange. */
typedef struct {
  int64_t calls, samples;
  double total_ms, pack_ms, exchange_ms, unpack_ms;
  int64_t pencil_bytes;
} cudecompExtTransposeTimings_t;
cudecompResult_t cudecompExtGetTransposeTimings(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, int32_t op,
                                                cudecompExtTransposeTimings_t* timings);
/* The same for cudecompUpdateHalos{X,Y,Z} (axis 0..2) along `dim`: pack / exchange / unpack of the plain sequence
 * (CUDECOMP_DISABLE_HALO_OVERLAP=1); the overlapped sequence reports the whole update as exchange time.  pencil_bytes
 * holds the bytes this rank put on the wire per update. */
cudecompResult_t cudecompExtGetHaloTimings(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, int32_t axis, int32_t dim,
                                           cudecompExtTransposeTimings_t* timings);

/* Peer-transport self test (collective): every rank writes tagged 4 KiB blocks at the start, middle, end and
 * every GiB boundary of the NEXT rank's copy of `buffer` (from cudecompMalloc, `bytes` long) and verifies what the
 * previous rank wrote into its own.  *mismatches = number of wrong blocks (0 = the IPC mapping is sound). */
cudecompResult_t cudecompExtPeerProbe(cudecompHandle_t handle, void* buffer, size_t bytes, int32_t* mismatches);

/* Which executor paths a descriptor's transposes have taken so far (tests assert that the intended path ran):
 * graphs_captured / graph_launches -- CUDECOMP_ENABLE_CUDA_GRAPHS: distinct pack loops captured, graph launches;
 * local -- no exchange; rccl, mpi -- those transports; peer_barrier -- one-sided exchange, all chunks at once (the
 * name dates from the host-barrier implementation; it is ordered by device-side flags now);
 * peer_fused -- fused pack+put (NVSHMEM_SM); peer_pipelined -- staged pipeline of the one-sided
 * transport (all peers in every stage). */
typedef struct {
  int64_t graphs_captured, graph_launches;
  int64_t local, rccl, mpi, peer_barrier, peer_fused, peer_pipelined;
  int64_t direct_puts; /* peer_fused transposes that wrote straight into the peers' output pencils (no unpack) */
  /* per HANDLE (not per descriptor): cudecompMalloc calls served from the pool of released workspaces, and new IPC
   * mappings whose page tags did not read back (stale mapping: the workspace was re-created at another address) */
  int64_t workspace_pool_hits, stale_ipc_mappings;
  /* per HANDLE: bytes cudecompFree has parked in the workspace pool right now; IPC mappings of re-created user buffers
   * that are kept open (the newest 32 survive a cudecompGridDescDestroy) */
  int64_t workspace_pool_bytes, retired_imports;
  /* user compute queues the kernel driver held on this process's GPU at the LAST census (when the one-sided transport
   * came up, at cudecompExtQueueCensus), over ALL processes (-1: none taken / not readable), and the hardware queue slots
   * that GPU has for them: more queues than slots = the driver time-slices every process of the device (ranks sharing a
   * GPU) */
  int64_t compute_queues_on_device, hardware_queue_slots;
  /* transposes of this descriptor whose exchange went through the two-hop relay (CUDECOMP_TWO_HOP_RELAY=1) */
  int64_t relayed;
  int64_t rotations;   /* single-rank in-place transposes run as one in-place rotation kernel */
} cudecompExtCounters_t;
cudecompResult_t cudecompExtGetCounters(cudecompHandle_t handle, cudecompGridDesc_t grid_desc,
                                        cudecompExtCounters_t* counters);

/* cudecompFree PARKS workspaces of the one-sided transports (allocation and peer mappings stay; the next cudecompMalloc
 * of a fitting size reuses them): up to min(1/8 of the device memory, 32 GiB) per handle -- the smallest such limit
 * over all ranks; CUDECOMP_WORKSPACE_POOL_MIB overrides it, 0 = park nothing -- until cudecompFinalize.  A
 * cudecompMalloc that runs out of memory releases the pool by itself and retries; an APPLICATION that needs the memory
 * for its own allocations calls this.  Collective over the handle's communicator. */
cudecompResult_t cudecompExtTrimWorkspacePool(cudecompHandle_t handle);

/* A fresh census of the compute queues all processes hold on this process's GPU and of its hardware queue slots (from the
 * kernel driver's tables; -1 / 0 if they cannot be read).  Local, but not free: do not call it per operation. */
cudecompResult_t cudecompExtQueueCensus(cudecompHandle_t handle, int32_t* compute_queues, int32_t* hardware_queue_slots);

/* One-direction copy rate from this rank to the next rank of the node, measured when the one-sided transport came up
 * (64 MiB, both engines, slowest rank): gbps_sdma through hipMemcpyAsync (copy engines), gbps_cu through the library's
 * copy kernel.  measured = 0: no measurement (one rank, no GPU, or CUDECOMP_SKIP_LINK_PROBE); crosses_devices = 0: the
 * ranks share one GPU (the figure is then a local copy rate, not a link rate); copy_engine: 0 copy engines, 1 kernel. */
typedef struct {
  double gbps_sdma, gbps_cu;
  int32_t measured, crosses_devices, copy_engine, reserved;
} cudecompExtLinkInfo_t;
cudecompResult_t cudecompExtGetLinkInfo(cudecompHandle_t handle, cudecompExtLinkInfo_t* info);

/* Name of the data-movement kernel this process launched last, as its template is spelled in csrc/kernels_*.hip (e.g.
 * "transpose_kernel<8,2,64,64,2,true>"); "" before the first launch.  The string is owned by the library. */
const char* cudecompExtLastKernelName(void);

/* The autotuner's analytic prior (csrc/autotune.cc estimateTransposeCycleMs): cost in ms of one X->Y->Z->Y->X cycle of
 * `es`-byte elements for the grid and transpose backend given, from the phases the plan really executes, the HBM rate
 * and one chunk per link (CUDECOMP_MODEL_HBM_GBPS / _XGMI_LINK_GBPS / _NIC_GBPS override the rates).  library_buffers:
 * pencils live in cudecompMalloc memory (NVSHMEM_SM then writes the destination pencils directly). */
cudecompResult_t cudecompExtEstimateCycleMs(cudecompHandle_t handle, const cudecompExtGridSpec_t* grid, int32_t es,
                                            int32_t backend, int32_t library_buffers, int32_t inplace, double* ms);

/* Run one block move on the GPU (src/dst are device pointers, strides in elements of es bytes).
 * force_generic is a bit mask: 1 selects the element-wise fallback kernel, 2 forces the streaming
 * (non-temporal) variants that are normally used only for moves of 32 MiB and more, 4 selects the window variant of
 * the LDS transpose for every destination off the 64-byte grid (normally only for moves of 1 MiB and more), 8 disables
 * it; 16 / 32: 128 x 64 / 64 x 64 tiles for 4-byte transposes (tuning variants; the default is 64 x 128); 64 / 128: transposes walk
 * their tiles i first / j first (without runs); 256: the move covers whole rows of a halo-carrying destination, the cells between
 * consecutive rows may be rewritten with their own content (dense row copy).  *kernel_class (optional) receives the
 * kernel flavour used: 0 rows, 1 LDS transpose, 2 generic. */
cudecompResult_t cudecompExtMove3D(const void* src, void* dst, int32_t es, const int64_t extent[3],
                                   const int64_t ss[3], const int64_t ds[3], int32_t force_generic,
                                   int32_t* kernel_class, hipStream_t stream);

/* How the kernel layer WOULD execute a 3-D block move between buffers at the given addresses (no launch; works without a
 * GPU): out[10] = {class (0 rows, 1 LDS transpose, 2 generic), kernel variant, tile_i, tile_j, tiles_i, tiles_j, batch extent,
 * run length of the tile walk, walk bits (1 XCD-contiguous, 2 j first, 4 runs over batch planes), access mode}.  flags: 2 =
 * streaming access regardless of the size, 4 = window / shifted variants regardless of the size, 64 / 128 = force the i-first /
 * j-first walk, 256 = whole destination rows (as cudecompExtMove3D), 8 = never rewrite the cells between rows (what
 * CUDECOMP_PRESERVE_OUTPUT_HALOS=1 does), flags >> 12 = the destination pencil's row pitch as the planner reports it
 * (cudecompExtMove_t::row_pitch; 0 = none).  Walk bit 8 = transpose_lines_kernel.  Row copies report their kernel in the
 * tile_i slot (0 plain, 1 shifted, 2 dense).  Harness-only (tests/test_kernel_plan.py). */
cudecompResult_t cudecompExtDescribeMove(uint64_t src_address, uint64_t dst_address, int32_t es, const int64_t extent[3],
                                        const int64_t ss[3], const int64_t ds[3], int32_t flags, int64_t out[10]);

/* The orbit walk of the in-place rotation kernel (csrc/rotate_walk.h; no launch, works without a GPU): for an array of nb
 * blocks per edge and walk (-1 = the default), *grid = the workgroups a launch has, and for workgroups first .. first + count - 1
 * blocks[3 * i .. 3 * i + 2] = the block triple (b0, b1, b2) of workgroup first + i, or -1 -1 -1 when it maps to none (padding).
 * blocks may be NULL (count 0).  Harness-only (tests/test_kernel_plan.py: every walk visits every triple exactly once). */
cudecompResult_t cudecompExtRotateWalk(int32_t nb, int32_t walk, int64_t first, int64_t count, int32_t* blocks, int64_t* grid);

#ifdef __cplusplus
}
#endif
#endif
