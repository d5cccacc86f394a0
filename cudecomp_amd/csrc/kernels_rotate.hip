// kernels_rotate.hip -- IN-PLACE axis rotation of a cubic array: the hops of an in-place X->Y->Z->Y->X cycle on a 1 x 1 grid in
// the all-axis-contiguous layout, one read and one write per element.  One code object of its own (kernels_batch.h).
//
// Replaces the staged single-rank in-place transpose -- permute into the workspace, copy back: two reads and two writes per
// element -- of the reference (include/internal/transpose.h:326-362) and of this library's generic plan (plan.cc) for the one
// shape where the permutation's cycles are short enough to be closed inside a workgroup.
#include "kernels_dev.h"

#include "errors.h"
#include "rotate_walk.h"

namespace cudecomp {
namespace kern {
namespace {

// An N x N x N array, memory position p = (p0, p1, p2), p0 fastest.  Going from the X pencil (x, y, z) to the Y pencil
// (y, z, x), from Y to Z (z, x, y) and back are, in place, the two rotations
//     forward:  new[p0, p1, p2] = old[p2, p0, p1]          inverse:  new[p0, p1, p2] = old[p1, p2, p0],
// permutations of the positions whose cycles have length 3 (1 on the diagonal).  Cut into T^3 tiles, the tile at block
// (b0, b1, b2) takes its new content from the tile at (b2, b0, b1) (forward), that one from (b1, b2, b0), that one from
// (b0, b1, b2): a workgroup OWNS one such orbit of three tiles (the owner is the lexicographically smallest of the three
// block triples; the other two workgroups of the orbit leave at once), loads all three tiles into registers, and only then
// stores them, each rotated, where they belong.  Orbits are disjoint, so no workgroup ever reads what another one writes.
//
// Inside a tile the rotation is a plain 2-D transposition of the tile's linear index space: with x the 16-valued and y the
// 256-valued coordinate,
//     forward:  old linear = x + T*y   (x = a0, y = a1 + T*a2)   ->  new linear = y + T^2*x
//     inverse:  old linear = y + T^2*x (y = a0 + T*a1, x = a2)   ->  new linear = x + T*y,
// staged through LDS as tile[x][y] with a row pitch of T^2 + 2 elements: the side whose vectors run along y moves whole
// 16-byte vectors, the side whose vectors run along x moves two elements a row apart -- 64 lanes x 8 bytes spread over all
// banks twice, which is the floor for 512 bytes per instruction.
//
// Granularity: a tile row is T elements = 128 bytes for 8-byte elements (one cache line), which is what the cubic tiles of an
// in-place rotation allow (a tile that is longer along p0 has an image that is longer along p1: no closed set of boxes).
template <int ES, int T, bool FWD>
__global__ __launch_bounds__(kThreads) void rotate_kernel(char* base, int n, int nb, int walk) {
  using E = Bytes<ES>;
  constexpr int VW = 16 / ES;               // elements per 16-byte vector
  using V = Bytes<ES * VW>;
  constexpr int TILE = T * T * T;
  constexpr int NV = TILE / (kThreads * VW);  // vectors per lane and tile
  constexpr int PY = T * T + 2;               // LDS row pitch (elements)
  static_assert(TILE % (kThreads * VW) == 0 && T % VW == 0, "rotate mapping");
  __shared__ __attribute__((aligned(16))) E tile[T * PY];

  // which orbit this workgroup takes: rotate_walk.h
  int b0, b1, b2;
  if (!rotateWalkBlock(blockIdx.x, (unsigned int)nb, walk, &b0, &b1, &b2)) return;
  // owner of the orbit {(b0,b1,b2), (b2,b0,b1), (b1,b2,b0)}: the lexicographically smallest triple (b2 most significant)
  const long long key0 = ((long long)b2 * nb + b1) * nb + b0, key1 = ((long long)b1 * nb + b0) * nb + b2, key2 = ((long long)b0 * nb + b2) * nb + b1;
  if (key0 > key1 || key0 > key2) return;
  const bool single = b0 == b1 && b1 == b2;

  E* const p = reinterpret_cast<E*>(base);
  const long long N = n, N2 = (long long)n * n;
  auto origin = [&](int c0, int c1, int c2) { return (long long)c0 * T + N * ((long long)c1 * T) + N2 * ((long long)c2 * T); };
  // D[0] = mine; its new content comes from S(D[0]); forward: S(b0,b1,b2) = (b2,b0,b1), inverse: S = (b1,b2,b0)
  long long org[3];
  org[0] = origin(b0, b1, b2);
  if (FWD) {
    org[1] = origin(b2, b0, b1);
    org[2] = origin(b1, b2, b0);
  } else {
    org[1] = origin(b1, b2, b0);
    org[2] = origin(b2, b0, b1);
  }
  // tile t = org[t]; new content of org[t] = rotated old content of org[(t + 1) % 3]
  const int tid = threadIdx.x;
  const int nt = single ? 1 : 3;

  // ---- all loads of the orbit first (registers), in the tiles' own linear order: lane vector v covers local linear
  //      lin = VW * (tid + kThreads * v) = c0 + T*c1 + T^2*c2
  V regs[3][NV];
  // (one straight-line block per case: all 3 * NV loads of an orbit are issued before anything waits)
  if (single) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int lin = VW * (tid + kThreads * v);
      const int c0 = lin % T, c1 = (lin / T) % T, c2 = lin / (T * T);
      regs[0][v] = loadVec<true, ES * VW>(p + org[0] + c0 + N * c1 + N2 * c2);
      regs[1][v] = regs[0][v];
      regs[2][v] = regs[0][v];
    }
  } else {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int lin = VW * (tid + kThreads * v);
        const int c0 = lin % T, c1 = (lin / T) % T, c2 = lin / (T * T);
        regs[t][v] = loadVec<true, ES * VW>(p + org[t] + c0 + N * c1 + N2 * c2);
      }
    }
  }
  // ---- tile by tile through LDS: old content of tile s = (t + 1) % 3 -> new content of tile t
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t < nt) {
      if (t > 0) __syncthreads();
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int lin = VW * (tid + kThreads * v);
        const V val = regs[(t + 1) % 3][v];  // (a diagonal tile holds its own content in all three sets)
        if constexpr (FWD) {  // old linear = x + T*y, vector along x: VW scalars a row apart
          const int x = lin % T, y = lin / T;
#pragma unroll
          for (int w = 0; w < VW; ++w) tile[(x + w) * PY + y] = Lane<ES, VW>::get(val, w);
        } else {              // old linear = y + T^2*x, vector along y: one 16-byte store
          const int y = lin % (T * T), x = lin / (T * T);
          *reinterpret_cast<V*>(tile + x * PY + y) = val;
        }
      }
      __syncthreads();
      E* const dst = p + org[t];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int lin = VW * (tid + kThreads * v);
        const int c0 = lin % T, c1 = (lin / T) % T, c2 = lin / (T * T);
        V out;
        if constexpr (FWD) {  // new linear = y + T^2*x, vector along y
          const int y = lin % (T * T), x = lin / (T * T);
          out = *reinterpret_cast<const V*>(tile + x * PY + y);
        } else {              // new linear = x + T*y, vector along x
          const int x = lin % T, y = lin / T;
#pragma unroll
          for (int w = 0; w < VW; ++w) Lane<ES, VW>::set(out, w, tile[(x + w) * PY + y]);
        }
        storeVec<ST_STREAM, ES * VW>(dst + c0 + N * c1 + N2 * c2, out);
      }
    }
  }
}

}  // namespace
}  // namespace kern

// tile edge per element size: 128-byte tile rows (16 fp64 / complex<fp32>, 8 complex<fp64>); 4-byte elements would get
// 64-byte rows (half lines) and keep the staged form
static int rotateTile(int es) { return es == 8 ? 16 : (es == 16 ? 8 : 0); }

bool rotateSupported(int es, long long n) {
  const int t = rotateTile(es);
  // (the grid bound holds for every walk: at most 2^5 - 1 blocks of padding per axis)
  return t > 0 && n >= t && n % t == 0 && rotateWalkGrid(n / t + 31, 0) < 0x7fffffffLL && n * n * n < (1ll << 40);
}

// direction: +1 forward (new[p0,p1,p2] = old[p2,p0,p1]), -1 inverse; buffer = the N^3 array (in place)
// walk: -1 = default, else as rotate_walk.h says (CUDECOMP_ROTATE_WALK in tuning builds)
void launchRotate(void* buffer, long long n, int es, int direction, hipStream_t stream, int walk) {
  if (!rotateSupported(es, n)) CD_INTERNAL_ERROR("in-place rotation not available for this shape");
  const int nb = (int)(n / rotateTile(es));
  walk = rotateWalkFor(walk, nb);
  const dim3 grid((unsigned int)rotateWalkGrid(nb, walk)), block(kern::kThreads);
  char* b = static_cast<char*>(buffer);
  if (es == 8) {
    if (direction > 0) kern::rotate_kernel<8, 16, true><<<grid, block, 0, stream>>>(b, (int)n, nb, walk);
    else kern::rotate_kernel<8, 16, false><<<grid, block, 0, stream>>>(b, (int)n, nb, walk);
  } else {
    if (direction > 0) kern::rotate_kernel<16, 8, true><<<grid, block, 0, stream>>>(b, (int)n, nb, walk);
    else kern::rotate_kernel<16, 8, false><<<grid, block, 0, stream>>>(b, (int)n, nb, walk);
  }
  CD_CHECK_HIP(hipGetLastError());
}

}  // namespace cudecomp
