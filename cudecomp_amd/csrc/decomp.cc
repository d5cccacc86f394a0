// decomp.cc -- see decomp.h
#include "decomp.h"

#include <algorithm>
#include <limits>

#include "errors.h"

namespace cudecomp {

std::array<int32_t, 2> gridIndexOfRank(const GridShape& g, int rank) {
  if (g.col_major) return {rank % g.pdims[0], rank / g.pdims[0]};
  return {rank / g.pdims[1], rank % g.pdims[1]};
}

int globalRankOf(const GridShape& g, const std::array<int32_t, 2>& pidx, CommAxis axis, int comm_rank) {
  // the member `comm_rank` of my row communicator shares my row index, etc.
  std::array<int32_t, 2> p = pidx;
  p[axis == COMM_ROW ? 1 : 0] = comm_rank;
  return g.col_major ? p[0] + p[1] * g.pdims[0] : p[0] * g.pdims[1] + p[1];
}

Pencil makePencil(const GridShape& g, const std::array<int32_t, 2>& pidx, int axis, const int32_t* halo,
                  const int32_t* pad) {
  constexpr i64 kI32Max = std::numeric_limits<int32_t>::max();
  Pencil p;
  int pos_of[3];
  for (int i = 0; i < 3; ++i) {
    p.order[i] = g.mem_order[axis][i];
    pos_of[p.order[i]] = i;
  }
  p.size = 1;
  int j = 0;  // grid dimension splitting the current global axis
  for (int ga = 0; ga < 3; ++ga) {
    const int m = pos_of[ga];
    i64 interior, lo = 0;
    if (ga == axis) {
      interior = g.gdims[ga];
    } else {
      const i64 n = g.gdims_dist[ga], np = g.pdims[j], me = pidx[j];
      const i64 base = n / np, rem = n % np;
      interior = base + (me < rem ? 1 : 0);
      // cells beyond gdims_dist belong to the last rank that owns anything
      if (me == std::min(np, n) - 1) interior += g.gdims[ga] - g.gdims_dist[ga];
      lo = me * base + std::min(me, rem);
      ++j;
    }
    if (interior < 0) CD_INVALID_USAGE("computed pencil shape values must be non-negative");
    if (interior > kI32Max) CD_INVALID_USAGE("computed pencil shape exceeds int32_t limit");
    p.lo[m] = (int32_t)lo;
    p.hi[m] = (int32_t)(lo + interior - 1);
    p.halo[ga] = halo ? halo[ga] : 0;
    p.pad[ga] = pad ? pad[ga] : 0;
    if (p.halo[ga] < 0) CD_INVALID_USAGE("halo_extents values must be non-negative");
    if (p.pad[ga] < 0) CD_INVALID_USAGE("padding values must be non-negative");
    const i64 full = interior + 2 * (i64)p.halo[ga] + p.pad[ga];
    if (full > kI32Max) CD_INVALID_USAGE("computed pencil shape exceeds int32_t limit");
    p.shape[m] = (int32_t)full;
    if (p.size == 0 || full == 0) {
      p.size = 0;
    } else {
      if (full > std::numeric_limits<i64>::max() / p.size) CD_INVALID_USAGE("computed pencil size exceeds int64_t limit");
      p.size *= full;
    }
  }
  return p;
}

std::vector<i64> splitExtent(i64 n, int nchunks, i64 surplus) {
  std::vector<i64> s(nchunks, n / nchunks);
  for (int i = 0; i < n % nchunks; ++i) s[i] += 1;
  s[std::min<i64>(n, nchunks) - 1] += surplus;
  return s;
}

std::vector<i64> prefixOffsets(const std::vector<i64>& splits) {
  std::vector<i64> off(splits.size(), 0);
  for (size_t i = 1; i < splits.size(); ++i) off[i] = off[i - 1] + splits[i - 1];
  return off;
}

bool anyEmptyPencil(const GridShape& g, int axis) {
  int j = 0;
  for (int ga = 0; ga < 3; ++ga) {
    if (ga == axis) continue;
    if (g.gdims_dist[ga] / g.pdims[j] == 0) return true;
    ++j;
  }
  return false;
}

i64 alignElements(i64 count) {
  // 256-byte granules counted in 4-byte units, independent of the dtype actually used
  return (count + 63) / 64 * 64;
}

i64 maxPencilElements(const GridShape& g, int axis) {
  i64 size = 1;
  int j = 0;
  for (int ga = 0; ga < 3; ++ga) {
    if (ga == axis) {
      size *= g.gdims[ga];
    } else {
      i64 d = (g.gdims_dist[ga] + g.pdims[j] - 1) / g.pdims[j];
      size *= d + (g.gdims[ga] - g.gdims_dist[ga]);
      ++j;
    }
  }
  return size;
}

i64 transposeWorkspaceElements(const GridShape& g) {
  const i64 x = maxPencilElements(g, 0), y = maxPencilElements(g, 1), z = maxPencilElements(g, 2);
  return std::max({alignElements(x) + y, alignElements(y) + x, alignElements(y) + z, alignElements(z) + y});
}

i64 haloWorkspaceElements(const GridShape& g, const std::array<int32_t, 2>& pidx, int axis, const int32_t* halo) {
  Pencil p = makePencil(g, pidx, axis, halo, nullptr);
  i64 best = 0;
  for (int d = 0; d < 3; ++d) {
    i64 face = p.extentG((d + 1) % 3) * p.extentG((d + 2) % 3) * p.halo[d];
    best = std::max(best, 4 * alignElements(face));
  }
  return best;
}

CommAxis commAxisOfDim(int axis, int dim) {
  // the first non-axis global dimension is split by pdims[0] (column comm), the second by pdims[1]
  for (int ga = 0; ga < 3; ++ga) {
    if (ga == axis) continue;
    return (ga == dim) ? COMM_COL : COMM_ROW;
  }
  return COMM_ROW;
}

int shiftedRank(const GridShape& g, int rank, int axis, int dim, int displacement, bool periodic) {
  if (displacement == 0) return rank;
  if (dim == axis) return periodic ? rank : -1;
  const CommAxis ca = commAxisOfDim(axis, dim);
  const auto pidx = gridIndexOfRank(g, rank);
  const int n = g.pdims[ca];
  const int shifted = pidx[ca == COMM_COL ? 0 : 1] + displacement;
  if (!periodic && (shifted < 0 || shifted >= n)) return -1;
  return globalRankOf(g, pidx, ca, (shifted + n) % n);
}

void alltoallPeers(int nranks, int npergroup, int rank, int iter, int* src_rank, int* dst_rank) {
  if (nranks == 1 || iter == 0) {
    *src_rank = *dst_rank = rank;
    return;
  }
  // odd steps walk the near half of the peers, even steps the far half, so near (intra-group)
  // and far (inter-group) transfers alternate
  const int step = (iter % 2 == 1) ? iter / 2 + 1 : nranks / 2 + iter / 2;
  if ((nranks & (nranks - 1)) == 0) {
    *src_rank = *dst_rank = rank ^ step;  // pairwise exchange: every step uses a distinct xGMI link
    return;
  }
  const int group = rank / npergroup, g0 = group * npergroup;
  if (step < npergroup) {
    *dst_rank = g0 + (rank + step) % npergroup;
    *src_rank = g0 + (rank + npergroup - step) % npergroup;
    return;
  }
  int d = (rank + step) % nranks;
  if (d >= g0 && d < g0 + npergroup) d = (d + npergroup) % nranks;
  int s = (rank + nranks - step) % nranks;
  if (s >= g0 && s < g0 + npergroup) s = (s + nranks - npergroup) % nranks;
  *dst_rank = d;
  *src_rank = s;
}

}  // namespace cudecomp
