// decomp.h -- host-only index math of the pencil decomposition: who owns what, where it lives in
// memory, who the neighbours are and how large the workspaces must be.  No HIP, no communication.
//
// Semantics follow NVIDIA/cuDecomp v0.7.0 so that results are interchangeable:
//   pencil geometry        reference src/cudecomp.cc:1317-1379
//   rank <-> grid index    reference include/internal/common.h:318-346
//   splits / alignment     reference include/internal/common.h:579-589, 632-640
//   workspace sizes        reference src/cudecomp.cc:1411-1459, include/internal/common.h:349-366
//   neighbour ranks        reference src/cudecomp.cc:1710-1755
//   all-to-all peer order  reference include/internal/common.h:533-577
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace cudecomp {

using i64 = int64_t;
using Int3 = std::array<int32_t, 3>;

enum CommAxis { COMM_COL = 0, COMM_ROW = 1 };  // column communicator spans pdims[0] ranks, row spans pdims[1]

// Resolved description of a decomposition (no defaults left to interpret).
struct GridShape {
  Int3 gdims{};
  Int3 gdims_dist{};
  std::array<int32_t, 2> pdims{};
  bool col_major = false;       // rank -> (row, col) assignment order
  int32_t mem_order[3][3] = {};  // [pencil axis][memory position] -> global axis
};

// Geometry of one rank's pencil.  shape/lo/hi/order are in MEMORY order, halo/pad in GLOBAL order
// (the public cudecompPencilInfo_t without its versioning header).
struct Pencil {
  Int3 shape{}, lo{}, hi{}, order{}, halo{}, pad{};
  i64 size = 0;

  // extent along GLOBAL axis g (including halos and padding)
  i64 extentG(int g) const {
    for (int i = 0; i < 3; ++i)
      if (order[i] == g) return shape[i];
    return 0;
  }
  // element stride of GLOBAL axis g in this pencil's memory
  i64 strideG(int g) const {
    if (order[0] == g) return 1;
    if (order[1] == g) return shape[0];
    return (i64)shape[0] * shape[1];
  }
  // element offset of the first interior cell
  i64 interiorOffset() const { return halo[0] * strideG(0) + halo[1] * strideG(1) + halo[2] * strideG(2); }
};

std::array<int32_t, 2> gridIndexOfRank(const GridShape& g, int rank);
int globalRankOf(const GridShape& g, const std::array<int32_t, 2>& pidx, CommAxis axis, int comm_rank);

// Pencil of the rank at grid position pidx.  Throws InvalidUsage on negative halos/padding or
// int32 / int64 overflow, like the reference.
Pencil makePencil(const GridShape& g, const std::array<int32_t, 2>& pidx, int axis, const int32_t* halo,
                  const int32_t* pad);

std::vector<i64> splitExtent(i64 n, int nchunks, i64 surplus);
std::vector<i64> prefixOffsets(const std::vector<i64>& splits);
bool anyEmptyPencil(const GridShape& g, int axis);

i64 alignElements(i64 count);  // round up so that count * 4 bytes is a multiple of 256 bytes
i64 maxPencilElements(const GridShape& g, int axis);
i64 transposeWorkspaceElements(const GridShape& g);
i64 haloWorkspaceElements(const GridShape& g, const std::array<int32_t, 2>& pidx, int axis, const int32_t* halo);

// communicator that exchanges pencil dimension `dim` of an axis-`axis` pencil
CommAxis commAxisOfDim(int axis, int dim);
int shiftedRank(const GridShape& g, int rank, int axis, int dim, int displacement, bool periodic);

// step `iter` (0 = self) of the pairwise all-to-all schedule for a communicator of nranks whose
// fast-interconnect groups hold npergroup consecutive ranks
void alltoallPeers(int nranks, int npergroup, int rank, int iter, int* src_rank, int* dst_rank);

}  // namespace cudecomp
