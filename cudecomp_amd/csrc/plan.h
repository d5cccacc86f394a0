// plan.h -- transposes and halo updates as DATA: a short list of strided block moves plus the
// exchange counts, built once per (operation, halos, padding, in-place, transport traits) and cached
// on the grid descriptor.  The executors (transpose.cc, halo.cc) only bind pointers and launch.
//
// Everything a transpose does locally is a Move3D:
//     dst[dst_off + k0*ds[0] + k1*ds[1] + k2*ds[2]] = src[src_off + k0*ss[0] + k1*ss[1] + k2*ss[2]]
// for 0 <= k_i < extent[i], all in ELEMENTS.  A pack, an unpack, a halo face copy and a full 3-D
// permutation are the same object; the kernel layer (kernels.cc + kernels_*.hip) picks the access pattern.
//
// What the plan must reproduce (bit-exact interior of the output pencil) is defined by
// NVIDIA/cuDecomp's cudecompTranspose_ / cudecompUpdateHalos_ (reference
// include/internal/transpose.h:196-905, include/internal/halo.h:41-315).  The plan is derived from
// the decomposition itself rather than transcribed from that code: see DESIGN.md "Plan".
#pragma once
#include <vector>

#include "decomp.h"

namespace cudecomp {

enum BufId : uint8_t { BUF_IN = 0, BUF_OUT = 1, BUF_WORK = 2 };

struct Move3D {
  BufId src_buf = BUF_IN, dst_buf = BUF_OUT;
  i64 src_off = 0, dst_off = 0;
  i64 extent[3] = {1, 1, 1};
  i64 ss[3] = {0, 0, 0};
  i64 ds[3] = {0, 0, 0};
  int peer = -1;  // communicator rank this move feeds / drains (-1: not tied to one peer)
  // > 0: the move writes WHOLE interior rows of the destination pencil (its unit-stride dim spans the pencil's interior
  // along the fastest memory axis) and this is the pencil's row pitch in elements.  The cells between the end of one row and
  // the start of the row one pitch further are then halo / padding cells of that pencil, written by nobody during the
  // operation, and the kernel layer may write whole cache lines across the row ends, putting back into those cells what it
  // read from them (rows_dense_kernel, kernels_rows.hip; transpose_lines_kernel, kernels_lines.hip; transpose_rowlines_kernel, kernels_rowlines.hip).  Only ever set for
  // local destinations (never for puts into a peer's pencil).  THE RULE this rests on: during a transpose nobody writes the
  // halo / padding cells of its output pencil -- no move of the plan (checked over random decompositions, tests/test_plan_sim.py),
  // no one-sided write of a peer (direct puts address interior cells, halo plans never target a peer's pencil: they exchange
  // through workspaces or whole contiguous faces, buildHaloPlan), and no kernel of the caller on another stream (the
  // documented contract; CUDECOMP_PRESERVE_OUTPUT_HALOS=1 for callers who cannot promise that).
  i64 dst_row_pitch = 0;

  i64 elements() const { return extent[0] * extent[1] * extent[2]; }
};

// Traits of the transport that change the plan (not the result).
struct TransportTraits {
  bool pipelined = false;        // per-peer overlap of pack / exchange / unpack
  bool symmetric_recv = false;   // one-sided peer writes: receive area must sit at the same workspace offset on
                                 // every rank and inside the workspace (never in the user's output buffer)
  bool self_exchange = false;    // test aid (CUDECOMP_TEST_SELF_EXCHANGE=1): a one-member communicator still runs
                                 // pack -> exchange (with itself) -> unpack, so that a single GPU drives the real
                                 // transports end to end
};

struct TransposePlan {
  // identification
  int ax_a = 0, ax_b = 0, ax_c = 0;
  CommAxis comm_axis = COMM_COL;
  int nranks = 1, comm_rank = 0;  // size of / my rank in the exchanging communicator
  bool noop = false;              // nothing to do at all (single rank, in place, identical layout)

  std::vector<Move3D> pack;    // before the exchange (schedule order: peers first, self last)
  std::vector<Move3D> unpack;  // after the exchange (self first)

  // exchange: chunk for member d starts at send_off[d] of (send_buf + send_base); the chunk from
  // member s lands at recv_off[s] of (recv_buf + recv_base); all in elements
  bool exchange = false;
  BufId send_buf = BUF_WORK, recv_buf = BUF_WORK;
  i64 send_base = 0, recv_base = 0;
  std::vector<i64> send_cnt, send_off, recv_cnt, recv_off;
  std::vector<i64> remote_recv_off;  // where MY chunk lands in member d's receive area (one-sided transports)
  std::vector<int> schedule_dst, schedule_src;  // pairwise peer order, entry 0 = self
  // Staged exchange (one-sided pipelined transports): every chunk is a dense block whose SLOWEST wire dim is the global
  // axis `stage_axis`; cutting all chunks into the same number of ranges along it gives sub-chunks that are contiguous
  // in the send and receive areas.  stage_limit = the smallest extent any chunk of the communicator has along that axis
  // (the same number on every member: an upper bound for the number of stages everybody can agree on without talking).
  int stage_axis = 2;
  i64 stage_limit = 1;
  i64 stage_elements = 0;  // largest pencil of the decomposition (the same number on every rank): staging is sized by it
  std::vector<i64> send_n, recv_n;  // extent along stage_axis of the chunk for member d / from member s

  // Direct-to-destination put (one-sided transports, out of place): move `direct[j]` takes the slab of my input that
  // belongs to member direct[j].peer and writes it straight into THAT member's output pencil, in its final layout
  // (dst_off / ds are relative to the peer's output buffer) -- one HBM pass per element, no receive area, no unpack.
  // Same order as `pack` (peers in schedule order, self last).  Empty when the plan has no such form (in place).
  std::vector<Move3D> direct;

  // Single-rank, in place, cubic, no halos / padding, and the two memory orders a rotation of each other: the whole
  // operation is the in-place rotation new[p0,p1,p2] = old[p2,p0,p1] (+1) or its inverse (-1) of an n^3 array -- one read
  // and one write per element where pack + unpack through the workspace (which stay in the plan: the kernel layer says
  // whether it has the rotation for the element size, kernels_rotate.hip) need two of each.  0: no such form.
  int rotate = 0;
  i64 rotate_n = 0;

  i64 pencil_elements_a = 0;  // interior elements moved (for bandwidth accounting)
};

enum TransposeOp { OP_X_TO_Y = 0, OP_Y_TO_Z = 1, OP_Z_TO_Y = 2, OP_Y_TO_X = 3 };

TransposePlan buildTransposePlan(const GridShape& g, int rank, TransposeOp op, const int32_t* in_halo,
                                 const int32_t* out_halo, const int32_t* in_pad, const int32_t* out_pad, bool inplace,
                                 const TransportTraits& traits, int npergroup);

// ---- two-hop relay of a low-fan-out exchange over the whole node (transport.cc: peerRelayAlltoall) ----------------------
// On a full xGMI mesh an exchange among P members drives P - 1 of a GPU's links.  On a pencil grid P is small -- the
// X<->Y exchange of a 2 x 4 grid has P = 2: half a pencil through ONE link while six links idle.  Here every outgoing chunk
// is cut into `nranks` equal slices (nranks = all ranks of the node); slice q travels source -> rank q -> destination, the
// slices q = source and q = destination go straight to the destination.  Every link then carries two slices per direction
// instead of one link carrying the whole chunk: wire time / (nranks / 2), paid with one extra HBM round trip of the relayed
// bytes at the relays.  (The grouping idea of the reference's schedule, include/internal/common.h:533-577, taken one step
// further; nothing like it exists there.)
//   step 1 "scatter": my slices -> relay regions of the ranks q (relay slot of (source, chunk index)) / receive area of the
//                     destination for the two direct slices;
//   step 2 "forward": what arrived in MY relay region -> the receive areas of its destinations.
// Every rank derives the moves of every other rank from the decomposition alone (the planner is stateless), so the
// forwarder knows where a slice must go without any metadata travelling with it.
struct RelayMove {
  int dst_rank = 0;      // GLOBAL rank whose memory is written
  bool to_relay = false; // destination is that rank's relay region (else its receive area)
  i64 src_off = 0;       // scatter: elements from the start of my send area; forward: elements into MY relay region
  i64 dst_off = 0;       // elements into the destination's relay region / receive area
  i64 count = 0;
};
struct RelayPlan {
  bool applies = false;
  int nranks = 0;            // ranks of the node (= of the handle)
  int slots_per_source = 0;  // chunks a rank sends = P - 1
  i64 slot_elements = 0;     // size of one relay slot = the largest slice of the decomposition
  std::vector<RelayMove> scatter, forward;
  i64 relayElements() const { return (i64)nranks * slots_per_source * slot_elements; }
};
// worth it when the exchange uses at most a third of the links a rank has
inline bool relayWorthwhile(int P, int nranks) { return P >= 2 && nranks >= 4 && 3 * (P - 1) <= nranks - 1; }
RelayPlan buildRelayPlan(const GridShape& g, int nranks, int rank, TransposeOp op, const int32_t* in_halo,
                         const int32_t* out_halo, const int32_t* in_pad, const int32_t* out_pad, bool inplace,
                         const TransportTraits& traits, int npergroup);

struct HaloPlan {
  int axis = 0, dim = 0;
  enum Kind { NONE, SELF_PERIODIC, PACKED, DIRECT } kind = NONE;
  CommAxis comm_axis = COMM_COL;
  int neighbor[2] = {-1, -1};  // global ranks of the -1 / +1 neighbours (-1: none)
  std::vector<Move3D> pre;     // SELF_PERIODIC: the two wrap copies; PACKED: face -> workspace
  std::vector<Move3D> post;    // PACKED: workspace -> halo
  // exchange in elements.  Face i (0 = low side, 1 = high side) is SENT to neighbour i and halo slot i is
  // FILLED by neighbour i.  Offsets are relative to the pencil (DIRECT) or the workspace (PACKED).
  i64 face_elements = 0;
  BufId xbuf = BUF_WORK;
  i64 send_off[2] = {0, 0}, recv_off[2] = {0, 0};
};

HaloPlan buildHaloPlan(const GridShape& g, int rank, int axis, int dim, const int32_t* halo, const bool* periods,
                       const int32_t* pad, bool force_packed, bool self_exchange = false);

// Number of stages every member of the communicator arrives at without talking: at most `wanted`, at most the smallest
// chunk extent, at most 14 (flag steps), and no stage smaller than `min_stage_bytes` of the largest pencil (below that the extra
// launches cost more than the overlap gains).
int stageCount(const TransposePlan& p, int wanted, int es, i64 min_stage_bytes = (i64)8 << 20);

// range k of K (equal parts, the remainder spread over the first ranges) of the extent of `m` along global axis `axis`
Move3D stageOfMove(const Move3D& m, int axis, int k, int K);

// canonical form used by the kernel layer: unit-extent dims dropped, mergeable dims fused, dims
// ordered by source stride.  Returns the number of remaining dims (0..3).
int normalizeMove(Move3D& m);

}  // namespace cudecomp
