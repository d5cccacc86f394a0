// plan.cc -- see plan.h.
#include "plan.h"

#include <algorithm>
#include <utility>

#include "errors.h"

namespace cudecomp {

namespace {

struct OpAxes {
  int a, b, c;
};

OpAxes axesOf(TransposeOp op) {
  // X->Y, Y->Z step "forward" through the axes, Z->Y, Y->X step back
  switch (op) {
    case OP_X_TO_Y: return {0, 1, 2};
    case OP_Y_TO_Z: return {1, 2, 0};
    case OP_Z_TO_Y: return {2, 1, 0};
    default: return {1, 0, 2};
  }
}

bool anySet(const int32_t* v) { return v && (v[0] != 0 || v[1] != 0 || v[2] != 0); }

// strides (indexed by GLOBAL axis) of a dense block with extents E (by global axis) stored in `order`
void denseStrides(const Int3& order, const i64 E[3], i64 st[3]) {
  st[order[0]] = 1;
  st[order[1]] = E[order[0]];
  st[order[2]] = E[order[0]] * E[order[1]];
}

Move3D blockMove(BufId sb, i64 soff, const i64 sst[3], BufId db, i64 doff, const i64 dst[3], const i64 E[3],
                 int peer) {
  Move3D m;
  m.src_buf = sb;
  m.dst_buf = db;
  m.src_off = soff;
  m.dst_off = doff;
  for (int g = 0; g < 3; ++g) {
    m.extent[g] = E[g];
    m.ss[g] = sst[g];
    m.ds[g] = dst[g];
  }
  m.peer = peer;
  return m;
}

}  // namespace

TransposePlan buildTransposePlan(const GridShape& g, int rank, TransposeOp op, const int32_t* in_halo,
                                 const int32_t* out_halo, const int32_t* in_pad, const int32_t* out_pad, bool inplace,
                                 const TransportTraits& traits, int npergroup) {
  TransposePlan p;
  const OpAxes ax = axesOf(op);
  p.ax_a = ax.a;
  p.ax_b = ax.b;
  p.ax_c = ax.c;
  // whichever of the two pencils involves Z is exchanged inside a row of the process grid
  p.comm_axis = (ax.a == 2 || ax.b == 2) ? COMM_ROW : COMM_COL;
  const int P = g.pdims[p.comm_axis == COMM_COL ? 0 : 1];
  const auto pidx = gridIndexOfRank(g, rank);
  const int me = pidx[p.comm_axis == COMM_COL ? 0 : 1];
  p.nranks = P;
  p.comm_rank = me;

  const Pencil a = makePencil(g, pidx, ax.a, nullptr, nullptr);
  const Pencil ah = makePencil(g, pidx, ax.a, in_halo, in_pad);
  const Pencil b = makePencil(g, pidx, ax.b, nullptr, nullptr);
  const Pencil bh = makePencil(g, pidx, ax.b, out_halo, out_pad);
  if (anyEmptyPencil(g, ax.a) || anyEmptyPencil(g, ax.b))
    CD_NOT_SUPPORTED("transposes on configurations with empty pencils not supported");

  const bool orders_equal = (a.order == b.order);
  const bool in_hp = anySet(in_halo) || anySet(in_pad);
  const bool out_hp = anySet(out_halo) || anySet(out_pad);
  bool hp_equal = true;
  for (int i = 0; i < 3; ++i) {
    if ((in_halo ? in_halo[i] : 0) != (out_halo ? out_halo[i] : 0)) hp_equal = false;
    if ((in_pad ? in_pad[i] : 0) != (out_pad ? out_pad[i] : 0)) hp_equal = false;
  }

  i64 Sa[3], Sb[3], ast[3], bst[3];
  for (int ga = 0; ga < 3; ++ga) {
    Sa[ga] = a.extentG(ga);
    Sb[ga] = b.extentG(ga);
    ast[ga] = ah.strideG(ga);
    bst[ga] = bh.strideG(ga);
  }
  p.pencil_elements_a = a.size;

  if (P == 1 && !traits.self_exchange) {
    // The whole transpose is local.  Out of place: one move straight from the input interior to the
    // output interior (a copy when the layouts agree, a permutation otherwise).  In place: nothing to
    // do if the layouts agree, else stage the interior through the workspace.
    if (!inplace) {
      p.pack.push_back(blockMove(BUF_IN, ah.interiorOffset(), ast, BUF_OUT, bh.interiorOffset(), bst, Sa, 0));
      p.pack.back().dst_row_pitch = Sb[b.order[0]] > 1 ? bst[b.order[1]] : 0;  // the whole interior of the output pencil
    } else if (orders_equal && hp_equal) {
      p.noop = true;
    } else {
      i64 wst[3];
      denseStrides(b.order, Sb, wst);
      p.pack.push_back(blockMove(BUF_IN, ah.interiorOffset(), ast, BUF_WORK, 0, wst, Sa, 0));
      p.unpack.push_back(blockMove(BUF_WORK, 0, wst, BUF_OUT, bh.interiorOffset(), bst, Sb, 0));
      p.unpack.back().dst_row_pitch = Sb[b.order[0]] > 1 ? bst[b.order[1]] : 0;
      // cubic and halo-free: memory position i of the input holds axis a.order[i]; the output wants b.order[i] there
      if (!in_hp && !out_hp && Sa[0] == Sa[1] && Sa[1] == Sa[2] && Sa[0] > 1) {
        if (a.order[0] == b.order[2] && a.order[1] == b.order[0] && a.order[2] == b.order[1]) p.rotate = 1;    // new[p] = old[p2,p0,p1]
        if (a.order[0] == b.order[1] && a.order[1] == b.order[2] && a.order[2] == b.order[0]) p.rotate = -1;   // new[p] = old[p1,p2,p0]
        if (p.rotate) p.rotate_n = Sa[0];
      }
    }
    p.schedule_dst = {0};
    p.schedule_src = {0};
    return p;
  }

  const auto splits_a = splitExtent(g.gdims_dist[ax.a], P, g.gdims[ax.a] - g.gdims_dist[ax.a]);
  const auto splits_b = splitExtent(g.gdims_dist[ax.b], P, g.gdims[ax.b] - g.gdims_dist[ax.b]);
  const auto off_a = prefixOffsets(splits_a);
  const auto off_b = prefixOffsets(splits_b);

  // What travels: member d gets the slab off_a[d] .. +splits_a[d] of my pencil along ax_a; from member s I
  // get the slab off_b[s] .. +splits_b[s] of my output pencil along ax_b.  Chunks are dense blocks stored
  // in a "wire order" W.  If ax_a is the slowest axis of the input (and it carries no halos/padding) the
  // chunks already sit in the input: send from there (W = input order).  If ax_b is the slowest axis of a
  // halo-free output the chunks can land in place: receive there (W = output order).  Per-peer overlapped
  // in-place operation keeps both stagings so a chunk never lands on data still to be sent.
  const bool elide_ok = !(traits.pipelined && inplace);
  const bool skip_pack = elide_ok && a.order[2] == ax.a && !in_hp;
  const bool skip_unpack = !skip_pack && elide_ok && !traits.symmetric_recv && b.order[2] == ax.b && !out_hp;
  Int3 W = a.order;
  if (!skip_pack && (skip_unpack || (b.order[2] == ax.a && !orders_equal))) W = b.order;

  p.exchange = true;
  // staging: all chunks are cut along their slowest wire dim.  Its extent is splits_a[d] (chunk for d) when that dim is
  // ax_a, the sender's slab splits_b[s] when it is ax_b, and the common extent along ax_c otherwise.
  p.stage_axis = W[2];
  if (W[2] == ax.a) p.stage_limit = *std::min_element(splits_a.begin(), splits_a.end());
  else if (W[2] == ax.b) p.stage_limit = *std::min_element(splits_b.begin(), splits_b.end());
  else p.stage_limit = Sa[ax.c];
  p.stage_elements = maxPencilElements(g, ax.a);
  p.send_buf = skip_pack ? BUF_IN : BUF_WORK;
  p.send_base = 0;
  p.recv_buf = skip_unpack ? BUF_OUT : BUF_WORK;
  if (skip_pack || skip_unpack) {
    p.recv_base = 0;
  } else {
    p.recv_base = alignElements(traits.symmetric_recv ? maxPencilElements(g, ax.a) : a.size);
  }

  p.send_cnt.resize(P);
  p.send_off.resize(P);
  p.recv_cnt.resize(P);
  p.recv_off.resize(P);
  p.remote_recv_off.resize(P);
  for (int i = 0; i < P; ++i) {
    p.send_cnt[i] = splits_a[i] * Sa[ax.b] * Sa[ax.c];
    p.send_off[i] = off_a[i] * Sa[ax.b] * Sa[ax.c];
    p.recv_cnt[i] = splits_b[i] * Sb[ax.a] * Sb[ax.c];
    p.recv_off[i] = off_b[i] * Sb[ax.a] * Sb[ax.c];
    p.remote_recv_off[i] = off_b[me] * splits_a[i] * Sa[ax.c];  // = member i's recv_off[me]
  }
  p.send_n.resize(P);
  p.recv_n.resize(P);
  for (int i = 0; i < P; ++i) {
    p.send_n[i] = (p.stage_axis == ax.a) ? splits_a[i] : Sa[p.stage_axis];
    p.recv_n[i] = (p.stage_axis == ax.b) ? splits_b[i] : Sb[p.stage_axis];
  }

  p.schedule_dst.resize(P);
  p.schedule_src.resize(P);
  for (int j = 0; j < P; ++j) alltoallPeers(P, npergroup, me, j, &p.schedule_src[j], &p.schedule_dst[j]);

  if (!skip_pack) {
    for (int j = 1; j <= P; ++j) {  // peers in schedule order, my own chunk last
      const int d = (j == P) ? me : p.schedule_dst[j];
      i64 E[3] = {Sa[0], Sa[1], Sa[2]}, wst[3];
      E[ax.a] = splits_a[d];
      denseStrides(W, E, wst);
      p.pack.push_back(blockMove(BUF_IN, ah.interiorOffset() + off_a[d] * ast[ax.a], ast, BUF_WORK,
                                 p.send_base + p.send_off[d], wst, E, d));
    }
  }
  if (traits.symmetric_recv && !inplace) {
    // the same slabs, written straight into the owners' output pencils (their geometry, their halos / padding)
    for (int j = 1; j <= P; ++j) {
      const int d = (j == P) ? me : p.schedule_dst[j];
      auto pidx_d = pidx;
      pidx_d[p.comm_axis == COMM_COL ? 0 : 1] = d;
      const Pencil bd = makePencil(g, pidx_d, ax.b, out_halo, out_pad);
      i64 E[3] = {Sa[0], Sa[1], Sa[2]}, dst[3];
      E[ax.a] = splits_a[d];
      for (int ga = 0; ga < 3; ++ga) dst[ga] = bd.strideG(ga);
      p.direct.push_back(blockMove(BUF_IN, ah.interiorOffset() + off_a[d] * ast[ax.a], ast, BUF_OUT,
                                   bd.interiorOffset() + off_b[me] * dst[ax.b], dst, E, d));
    }
  }
  if (!skip_unpack) {
    for (int j = 0; j < P; ++j) {  // my own chunk first, then peers in schedule order
      const int s = (j == 0) ? me : p.schedule_src[j];
      i64 E[3] = {Sb[0], Sb[1], Sb[2]}, wst[3];
      E[ax.b] = splits_b[s];
      denseStrides(W, E, wst);
      p.unpack.push_back(blockMove(BUF_WORK, p.recv_base + p.recv_off[s], wst, BUF_OUT,
                                   bh.interiorOffset() + off_b[s] * bst[ax.b], bst, E, s));
      // the chunks are slabs along ax_b: unless that is the output's fastest memory axis every chunk holds whole rows
      // (one-element rows excepted: the next axis is then the contiguous one, and it may be the split one)
      p.unpack.back().dst_row_pitch = ((b.order[0] != ax.b) && Sb[b.order[0]] > 1) ? bst[b.order[1]] : 0;
    }
  }
  return p;
}

int stageCount(const TransposePlan& p, int wanted, int es, i64 min_stage_bytes) {
  const i64 by_size = min_stage_bytes > 0 ? std::max<i64>(1, p.stage_elements * es / min_stage_bytes) : (i64)wanted;
  return (int)std::max<i64>(1, std::min<i64>({(i64)wanted, p.stage_limit, (i64)14, by_size}));
}

Move3D stageOfMove(const Move3D& m, int axis, int k, int K) {
  Move3D r = m;
  const i64 n = m.extent[axis], lo = n * k / K, hi = n * (k + 1) / K;
  r.extent[axis] = hi - lo;
  r.src_off += lo * m.ss[axis];
  r.dst_off += lo * m.ds[axis];
  if (K > 1 && m.ds[axis] == 1) r.dst_row_pitch = 0;  // (a stage that cuts the rows themselves)
  return r;
}

HaloPlan buildHaloPlan(const GridShape& g, int rank, int axis, int dim, const int32_t* halo, const bool* periods,
                       const int32_t* pad, bool force_packed, bool self_exchange) {
  HaloPlan p;
  p.axis = axis;
  p.dim = dim;
  const auto pidx = gridIndexOfRank(g, rank);
  const Pencil h = makePencil(g, pidx, axis, halo, nullptr);  // extents of what is exchanged
  const Pencil hp = makePencil(g, pidx, axis, halo, pad);     // strides of the user's buffer
  if (anyEmptyPencil(g, axis)) CD_NOT_SUPPORTED("halo operations on configurations with empty pencils not supported");

  const bool periodic = periods && periods[dim];
  p.neighbor[0] = shiftedRank(g, rank, axis, dim, -1, periodic);
  p.neighbor[1] = shiftedRank(g, rank, axis, dim, +1, periodic);
  const i64 he = halo[dim];
  if (he == 0) return p;

  p.comm_axis = commAxisOfDim(axis, dim);
  if (p.neighbor[0] == rank && p.neighbor[1] == rank && !self_exchange) {
    p.kind = HaloPlan::SELF_PERIODIC;
  } else if (p.neighbor[0] == -1 && p.neighbor[1] == -1) {
    return p;  // one rank along a non-periodic dimension
  } else {
    // only nearest-neighbour halos: the halo may not be wider than my slab or a neighbour's slab
    if (p.neighbor[0] == rank && p.neighbor[1] == rank) {  // self_exchange: periodic wrap onto myself, through the transport
      if (he > h.extentG(dim) - 2 * he) CD_INVALID_USAGE("halo wider than the pencil it wraps around");
    } else {
      const int np = g.pdims[p.comm_axis];
      const auto splits = splitExtent(g.gdims_dist[dim], np, g.gdims[dim] - g.gdims_dist[dim]);
      const int me = pidx[p.comm_axis == COMM_COL ? 0 : 1];
      int l = me - 1, r = me + 1;
      if (periodic) {
        l = (l + np) % np;
        r = (r + np) % np;
      }
      if ((l >= 0 && (he > splits[l] || he > splits[me])) || (r < np && (he > splits[r] || he > splits[me])))
        CD_INVALID_USAGE("halo includes ranks other than nearest neighbor processes, this is not currently supported.");
    }
    const bool faces_contiguous = (dim == h.order[2]);
    p.kind = (faces_contiguous && !anySet(pad) && !force_packed) ? HaloPlan::DIRECT : HaloPlan::PACKED;
  }

  // A face is the slab of thickness he along `dim`, spanning the other two dims INCLUDING their halos
  // (so that updating dims 0,1,2 in turn also fills edges and corners) but not their padding.
  i64 E[3], st[3], fst[3];
  for (int ga = 0; ga < 3; ++ga) {
    E[ga] = (ga == dim) ? he : h.extentG(ga);
    st[ga] = hp.strideG(ga);
  }
  denseStrides(h.order, E, fst);
  p.face_elements = E[0] * E[1] * E[2];
  const i64 sd = st[dim];
  const i64 n = hp.extentG(dim) - (pad ? pad[dim] : 0);  // extent along dim without padding
  const i64 lo_halo = 0, lo_face = he * sd, hi_face = (n - 2 * he) * sd, hi_halo = (n - he) * sd;

  switch (p.kind) {
    case HaloPlan::SELF_PERIODIC:
      p.pre.push_back(blockMove(BUF_IN, hi_face, st, BUF_IN, lo_halo, st, E, -1));
      p.pre.push_back(blockMove(BUF_IN, lo_face, st, BUF_IN, hi_halo, st, E, -1));
      break;
    case HaloPlan::PACKED: {
      const i64 A = alignElements(p.face_elements);
      p.xbuf = BUF_WORK;
      p.send_off[0] = 0;
      p.send_off[1] = A;
      p.recv_off[0] = 2 * A;
      p.recv_off[1] = 3 * A;
      if (p.neighbor[0] != -1) {
        p.pre.push_back(blockMove(BUF_IN, lo_face, st, BUF_WORK, p.send_off[0], fst, E, 0));
        p.post.push_back(blockMove(BUF_WORK, p.recv_off[0], fst, BUF_IN, lo_halo, st, E, 0));
      }
      if (p.neighbor[1] != -1) {
        p.pre.push_back(blockMove(BUF_IN, hi_face, st, BUF_WORK, p.send_off[1], fst, E, 1));
        p.post.push_back(blockMove(BUF_WORK, p.recv_off[1], fst, BUF_IN, hi_halo, st, E, 1));
      }
    } break;
    case HaloPlan::DIRECT:
      p.xbuf = BUF_IN;
      p.send_off[0] = lo_face;
      p.send_off[1] = hi_face;
      p.recv_off[0] = lo_halo;
      p.recv_off[1] = hi_halo;
      break;
    default: break;
  }
  return p;
}

int normalizeMove(Move3D& m) {
  struct D {
    i64 e, s, d;
  };
  D dims[3];
  int n = 0;
  for (int i = 0; i < 3; ++i)
    if (m.extent[i] != 1) dims[n++] = {m.extent[i], m.ss[i], m.ds[i]};
  std::sort(dims, dims + n, [](const D& x, const D& y) { return x.s != y.s ? x.s < y.s : x.d < y.d; });
  // fuse dims that are contiguous continuations of one another on BOTH sides
  bool fused = true;
  while (fused && n > 1) {
    fused = false;
    for (int i = 0; i < n && !fused; ++i)
      for (int j = 0; j < n && !fused; ++j) {
        if (i == j) continue;
        if (dims[j].s == dims[i].s * dims[i].e && dims[j].d == dims[i].d * dims[i].e) {
          dims[i].e *= dims[j].e;
          for (int k = j; k + 1 < n; ++k) dims[k] = dims[k + 1];
          --n;
          fused = true;
        }
      }
  }
  std::sort(dims, dims + n, [](const D& x, const D& y) { return x.s != y.s ? x.s < y.s : x.d < y.d; });
  for (int i = 0; i < 3; ++i) {
    if (i < n) {
      m.extent[i] = dims[i].e;
      m.ss[i] = dims[i].s;
      m.ds[i] = dims[i].d;
    } else {
      m.extent[i] = 1;
      m.ss[i] = 0;
      m.ds[i] = 0;
    }
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// two-hop relay (plan.h)
// ---------------------------------------------------------------------------------------------------------------
RelayPlan buildRelayPlan(const GridShape& g, int nranks, int rank, TransposeOp op, const int32_t* in_halo,
                         const int32_t* out_halo, const int32_t* in_pad, const int32_t* out_pad, bool inplace,
                         const TransportTraits& traits, int npergroup) {
  RelayPlan rp;
  rp.nranks = nranks;
  if (nranks != g.pdims[0] * g.pdims[1]) return rp;
  // the plans of all ranks (cheap: index math only)
  std::vector<TransposePlan> plans;
  plans.reserve(nranks);
  for (int s = 0; s < nranks; ++s) plans.push_back(buildTransposePlan(g, s, op, in_halo, out_halo, in_pad, out_pad, inplace, traits, npergroup));
  const TransposePlan& mine = plans[rank];
  if (!mine.exchange || !relayWorthwhile(mine.nranks, nranks)) return rp;
  const int P = mine.nranks;
  rp.slots_per_source = P - 1;
  for (const TransposePlan& p : plans)
    for (int i = 0; i < P; ++i)
      if (i != p.comm_rank) rp.slot_elements = std::max(rp.slot_elements, (p.send_cnt[i] + nranks - 1) / nranks);
  rp.slot_elements = alignElements(rp.slot_elements);  // slots start on 256-byte boundaries
  for (int s = 0; s < nranks; ++s) {
    const TransposePlan& p = plans[s];
    const auto pidx = gridIndexOfRank(g, s);
    int j = 0;  // index among the chunks s sends
    for (int i = 0; i < P; ++i) {
      if (i == p.comm_rank) continue;
      const int d = globalRankOf(g, pidx, p.comm_axis, i);
      const i64 cnt = p.send_cnt[i];
      for (int q = 0; q < nranks; ++q) {
        const i64 lo = cnt * q / nranks, hi = cnt * (q + 1) / nranks;
        if (hi == lo) continue;
        const i64 slot = ((i64)s * (P - 1) + j) * rp.slot_elements;
        if (q == s || q == d) {  // straight to the destination
          if (s == rank) rp.scatter.push_back({d, false, p.send_off[i] + lo, p.remote_recv_off[i] + lo, hi - lo});
        } else {
          if (s == rank) rp.scatter.push_back({q, true, p.send_off[i] + lo, slot, hi - lo});
          if (q == rank) rp.forward.push_back({d, false, slot, p.remote_recv_off[i] + lo, hi - lo});
        }
      }
      ++j;
    }
  }
  rp.applies = true;
  return rp;
}

}  // namespace cudecomp
