// kernels_window.hip -- the LDS-tiled transposition for DESTINATION rows off the 64-byte grid (halo-shifted pencils, odd row
// pitches): one code object (see kernels_dev.h for why there are several).
#include "kernels_dev.h"

#include "errors.h"

namespace cudecomp {
namespace kern {
namespace {

// ---------------------------------------------------------------------------------------------
// transpose_window_kernel: the same permutation for DESTINATION rows that do not start on 64-byte boundaries
// (halo-shifted pencils, odd row pitches).  A rectangular tile would write, for every destination row, a segment that
// begins and ends inside a 64-byte unit; those partial units reach HBM as partial writes and cost 15-20 % of the
// kernel (tuning notes: profiles/r02_tuning.md -- aligning the 16-byte stores alone does not help, the partial units
// themselves are the cost, and 64 bytes is the granularity that matters).  Here the tile of destination row i covers
//     j in [bj*TJ - p_i, bj*TJ - p_i + TJ),   p_i = element phase of the row's start inside a 64-byte unit,
// so every store of the body is a whole, aligned unit and only the two ends of each ROW (not of each tile) are
// partial.  Rows of one tile have different phases (the pitch is not a multiple of 64 bytes), so the tile loads the
// TJ + U - 1 source rows its windows can touch; the U - 1 extra rows are the previous tile's and hit in L2.  LDS is
// accessed element-wise here (row pitch TI + 1: the column reads of the store phase spread over the banks).
// e = {ei, ej, ek}; ss = {1, sj, sk}; ds = {di, 1, dk} (elements), as for transpose_kernel; t1 counts windows.
// ---------------------------------------------------------------------------------------------
template <int ES, int VW, int TI, int TJ, int STREAM, int NT = kThreads>
__global__ __launch_bounds__(NT) void transpose_window_kernel(const Batch b) {
  using E = Bytes<ES>;
  using V = Bytes<ES * VW>;
  constexpr int U = 64 / ES;            // elements per 64-byte unit
  constexpr int ROWS = TJ + U - 1;      // source rows a tile's windows can touch
  constexpr int PITCH = TI + 1;
  constexpr int TPR = TI / VW;          // lanes per source row segment
  constexpr int RPP = NT / TPR;   // source rows per load pass
  constexpr int NP = (ROWS + RPP - 1) / RPP;
  constexpr int TPO = TJ / VW;          // lanes per destination row window
  constexpr int RPO = NT / TPO;   // destination rows per store pass
  constexpr int NPO = TI / RPO;
  static_assert(NT % TPR == 0 && NT % TPO == 0 && TI % RPO == 0, "window mapping");
  __shared__ __attribute__((aligned(16))) E tile[ROWS * PITCH];

  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const unsigned int ti_n = b.t0[mi], tj_n = b.t1[mi];
  const unsigned int nb = b.first_block[mi + 1] - b.first_block[mi];
  unsigned int lt = lb;
  if (b.p1[mi] & 1) {  // XCD-contiguous walk, see transpose_kernel
    const unsigned int per = nb >> 3;
    if (lb < (per << 3)) lt = (lb & 7u) * per + (lb >> 3);
  }
  unsigned int bi, bj, rest;
  if (b.p1[mi] & 2) {
    bj = lt % tj_n;
    rest = lt / tj_n;
    bi = rest % ti_n;
    rest /= ti_n;
  } else {
    bi = lt % ti_n;
    rest = lt / ti_n;
    bj = rest % tj_n;
    rest /= tj_n;
  }
  const long long k = rest;
  const long long i0 = (long long)bi * TI, jb = (long long)bj * TJ - (U - 1);  // LDS row 0 holds source row jb
  const long long ei = m.e[0], ej = m.e[1];
  const long long sj = m.ss[1], di = m.ds[0];
  const E* __restrict__ src = reinterpret_cast<const E*>(m.src) + k * m.ss[2];
  E* __restrict__ dst = reinterpret_cast<E*>(m.dst) + k * m.ds[2];
  const int tid = threadIdx.x;
  const bool interior = i0 + TI <= ei && jb >= 0 && jb + ROWS <= ej;

  // ---- global -> registers (all loads issued before the first use) -> LDS, rows along i
  {
    const int li = (tid % TPR) * VW, lj = tid / TPR;
    const E* base = src + (jb + lj) * sj + i0 + li;
    V regs[NP] = {};
    if (interior) {
      // (its own copy of the loop: with the edge test of the other branch folded in, the compiler waited for single loads
      // between the passes; here all NP loads of a lane are in flight before the first use)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        if (jj < ROWS) regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(base + (long long)(p * RPP) * sj);
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        const long long j = jb + jj;
        if (jj < ROWS && i0 + li < ei && j >= 0 && j < ej)
          regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(base + (long long)(p * RPP) * sj);
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int jj = lj + p * RPP;
      if (jj < ROWS) {
        E* row = tile + jj * PITCH + li;
#pragma unroll
        for (int v = 0; v < VW; ++v) row[v] = Lane<ES, VW>::get(regs[p], v);
      }
    }
  }
  __syncthreads();
  // ---- LDS -> registers -> global: destination row i takes LDS rows (U-1) - p_i ... + TJ
  {
    const int c = tid % TPO, lr = tid / TPO;
    const unsigned long long dbase = (unsigned long long)(reinterpret_cast<uintptr_t>(dst)) / ES;
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
      const int ii = lr + p * RPO;
      const long long i = i0 + ii;
      if (!interior && i >= ei) continue;
      const int ph = (int)((dbase + (unsigned long long)(i * di)) & (unsigned long long)(U - 1));
      const int r = (U - 1) - ph + VW * c;  // LDS row of the lane's first element
      const long long j = jb + r;
      E* q = dst + i * di + j;
      V out;
#pragma unroll
      for (int v = 0; v < VW; ++v) Lane<ES, VW>::set(out, v, tile[(r + v) * PITCH + ii]);
      if (interior || (j >= 0 && j + VW <= ej)) {
        storeVec<storePolicyOf<STREAM>(), ES * VW>(q, out);
      } else {
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (j + v >= 0 && j + v < ej) storeVec<storePolicyOf<STREAM>(), ES>(q + v, Lane<ES, VW>::get(out, v));
      }
    }
  }
  if constexpr (STREAM == 3) remoteStoresDone();
}

template <int STREAM>
void launchWindowT(int variant, int es, const Batch& b, unsigned int blocks, hipStream_t stream, bool wide) {
  const dim3 grid(blocks), block(kThreads);
  if (es == 4) {
    if (variant == 4) transpose_window_kernel<4, 4, 64, 128, STREAM><<<grid, block, 0, stream>>>(b);
    else transpose_window_kernel<4, 1, 64, 128, STREAM><<<grid, block, 0, stream>>>(b);
  } else if (es == 8) {
#ifdef CUDECOMP_TUNING_VARIANTS
    if (variant == 2 && wide) transpose_window_kernel<8, 2, 128, 64, STREAM, 512><<<grid, dim3(512), 0, stream>>>(b);
    else
#endif
    if (variant == 2) transpose_window_kernel<8, 2, 64, 64, STREAM><<<grid, block, 0, stream>>>(b);
    else transpose_window_kernel<8, 1, 64, 64, STREAM><<<grid, block, 0, stream>>>(b);
  } else {
    transpose_window_kernel<16, 1, 32, 32, STREAM><<<grid, block, 0, stream>>>(b);
  }
  (void)wide;
  CD_CHECK_HIP(hipGetLastError());
}

}  // namespace
}  // namespace kern

void launchWindowBatch(int es, int variant, bool wide, int stream_access, const kern::Batch& b, unsigned int blocks, hipStream_t stream) {
  if (stream_access == 3) kern::launchWindowT<3>(variant, es, b, blocks, stream, wide);
  else if (stream_access == 4 || stream_access == 2) kern::launchWindowT<4>(variant, es, b, blocks, stream, wide);
  else kern::launchWindowT<0>(variant, es, b, blocks, stream, wide);
}

}  // namespace cudecomp
