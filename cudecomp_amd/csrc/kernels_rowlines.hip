// kernels_rowlines.hip -- the LDS-tiled transposition onto a halo-carrying pencil whose CONSECUTIVE DESTINATION ROWS are the
// tile's own rows (the inverse hops of an axis-contiguous cycle, and unpack-side permutations, onto pencils with halos /
// padding on their fastest axis): whole cache lines across the row ends.  One code object of its own (kernels_batch.h).
//
// Replaces, for this shape, cutensorPermute writing rows of the output pencil (reference include/internal/transpose.h:80-157,
// 651-681, 759-815).
#include "kernels_dev.h"

#include "errors.h"

namespace cudecomp {
namespace kern {
namespace {

// ---------------------------------------------------------------------------------------------
// transpose_rowlines_kernel.  Dims (i, j, k) as for transpose_kernel: i unit-stride in the source, j unit-stride in the
// destination, k the batch dim; e = {ei, ej, ek}; ss = {1, sj, sk}; ds = {di, 1, dk}.  Here di is the destination pencil's ROW
// PITCH: row i + 1 follows row i in memory after g = di - ej halo / padding cells, planes k are far apart.
// transpose_window_kernel writes such rows in whole 64-byte units, but every row still begins and ends inside a 128-byte line
// (a partly written line costs about four line times, profiles/r05_tuning.md section 3).  The other part of the line at the
// END of row i is the gap and the HEAD of row i + 1 -- the same tile column of the next row.
//
// So a row's last window simply runs on: the window of row i covers the local positions
//     l in [w * TJ - p_i, w * TJ - p_i + TJ),   p_i = element phase of the row's first cell inside a UB-byte unit,
// of the row's EXTENDED linear space: l < ej its own cells, ej <= l < di the gap cells (read from the destination and written
// back unchanged -- halo / padding cells of the output pencil nobody else writes during the operation: Move3D::dst_row_pitch,
// the contract of rows_dense_kernel), di <= l the head of row i + 1 (cells (i + 1, l - di)).  Row i's windows end with the
// window that contains its last gap cell; row i + 1 starts where that window ended (O_i = its overhang, a whole number of
// units), the last row of a plane ends at its last cell.  Every store of the body is a whole aligned vector, every window whole
// lines; per plane only the first row's head and the last row's tail are partial.
// LDS row r of a tile holds local position lb0 + r for the TI rows of the tile: a source row (TI elements along i; shifted by
// one element when the position belongs to the next row), or TI gathered gap cells.
// t0 = tiles along i, t1 = windows per row; p1 bit 1 = XCD-contiguous walk (windows first, then tile rows, then planes).
// ---------------------------------------------------------------------------------------------
template <int ES, int VW, int TI, int TJ, int STREAM, int UB>
__global__ __launch_bounds__(kThreads) void transpose_rowlines_kernel(const Batch b) {
  using E = Bytes<ES>;
  using V = Bytes<ES * VW>;
  constexpr int U = UB / ES;            // elements per alignment unit
  constexpr int ROWS = TJ + U - 1;      // local positions a tile's windows can touch
  constexpr int PITCH = TI + 1;
  constexpr int TPR = TI / VW;          // lanes per source row segment
  constexpr int RPP = kThreads / TPR;   // LDS rows per load pass
  constexpr int NP = (ROWS + RPP - 1) / RPP;
  constexpr int TPO = TJ / VW;          // lanes per destination window
  constexpr int RPO = kThreads / TPO;   // destination rows per store pass
  constexpr int NPO = TI / RPO;
  static_assert(kThreads % TPR == 0 && kThreads % TPO == 0 && TI % RPO == 0 && TJ % U == 0 && NP <= 64, "rowlines mapping");
  __shared__ __attribute__((aligned(16))) E tile[ROWS * PITCH];

  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const unsigned int ti_n = b.t0[mi], tw_n = b.t1[mi];
  const unsigned int nb = b.first_block[mi + 1] - b.first_block[mi];
  unsigned int lt = lb;
  if (b.p1[mi] & 1) {  // XCD-contiguous walk, see transpose_kernel
    const unsigned int per = nb >> 3;
    if (lb < (per << 3)) lt = (lb & 7u) * per + (lb >> 3);
  }
  const unsigned int w = lt % tw_n;
  unsigned int rest = lt / tw_n;
  const unsigned int bi = rest % ti_n;
  const long long k = rest / ti_n;

  const int ei = (int)m.e[0], ej = (int)m.e[1];
  const int di = (int)m.ds[0];
  const long long sj = m.ss[1];
  const int i0 = (int)bi * TI, lb0 = (int)w * TJ - (U - 1);  // LDS row 0 holds local position lb0
  const E* __restrict__ src = reinterpret_cast<const E*>(m.src) + k * m.ss[2];
  E* dst = reinterpret_cast<E*>(m.dst) + k * m.ds[2];  // (read for the gap cells: no __restrict__)
  const int tid = threadIdx.x;
  // kind of tile: 2 = every column exists and so does the row after the tile's last one (the shifted columns): no edge tests at
  // all; 1 = the tile ends with the plane's last row (only the last shifted column is missing); 0 = ragged
  const int kind = i0 + TI < ei ? 2 : (i0 + TI == ei ? 1 : 0);

  // ---- global -> registers (all loads issued before the first use) -> LDS
  {
    const int li = (tid % TPR) * VW, lj = tid / TPR;
    V regs[NP] = {};
    unsigned long long gap_passes = 0;  // passes in which this lane's LDS row is a gap row
    if (kind == 2) {
      // no per-lane control flow around the loads: a gap row loads the row's last own cell instead (replaced below), a
      // position below the row's start (first window) the first cell (never stored)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        const int l = lb0 + jj;
        const bool gap = l >= ej && l < di;
        const int j = l >= di ? l - di : (gap ? ej - 1 : (l < 0 ? 0 : l));
        const int shift = l >= di ? 1 : 0;
        if (jj < ROWS) {
          regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(src + (long long)j * sj + i0 + li + shift);
          if (gap) gap_passes |= 1ull << p;
        }
      }
    } else if (kind == 1) {
      // the same; the lanes on the tile's last columns take the shifted rows element by element (column ei does not exist;
      // what would come from it belongs to the overhang of the plane's last row, which is never stored)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        const int l = lb0 + jj;
        const bool gap = l >= ej && l < di;
        const int j = l >= di ? l - di : (gap ? ej - 1 : (l < 0 ? 0 : l));
        const int shift = l >= di ? 1 : 0;
        if (jj < ROWS) {
          if (shift && li + VW == TI) {
#pragma unroll
            for (int v = 0; v + 1 < VW; ++v)
              Lane<ES, VW>::set(regs[p], v, loadVec<loadsStream<STREAM>(), ES>(src + (long long)j * sj + i0 + li + v + 1));
          } else {
            regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(src + (long long)j * sj + i0 + li + shift);
          }
          if (gap) gap_passes |= 1ull << p;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        const int l = lb0 + jj;
        if (jj >= ROWS || l < 0) continue;
        if (l >= ej && l < di) {
          gap_passes |= 1ull << p;
          continue;
        }
        const int j = l >= di ? l - di : l;
        const int shift = l >= di ? 1 : 0;
        if (j >= ej) continue;  // (beyond anything a window of this tile stores)
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (i0 + li + v + shift < ei)
            Lane<ES, VW>::set(regs[p], v, loadVec<loadsStream<STREAM>(), ES>(src + (long long)j * sj + i0 + li + v + shift));
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
    // gap rows: the gap cell of every row of the tile -- what the destination holds there goes back unchanged
    if (gap_passes) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (gap_passes >> p & 1ull) {
          const int jj = lj + p * RPP;
          const int l = lb0 + jj;
          E* row = tile + jj * PITCH + li;
#pragma unroll
          for (int v = 0; v < VW; ++v)
            if (kind != 0 || i0 + li + v < ei) row[v] = loadVec<false, ES>(dst + (long long)(i0 + li + v) * di + l);
        }
      }
    }
  }
  __syncthreads();
  // ---- LDS -> registers -> global: row i takes LDS rows (U-1) - p_i ... + TJ of its extended linear space
  {
    const int c = tid % TPO, lr = tid / TPO;
    const unsigned long long dbase = (unsigned long long)(reinterpret_cast<uintptr_t>(dst)) / ES;
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
      const int ii = lr + p * RPO;
      const int i = i0 + ii;
      if (i >= ei) continue;
      const int ph = (int)((dbase + (unsigned long long)((long long)i * di)) & (unsigned long long)(U - 1));
      // what row i stores: from where row i - 1's last window ended (its overhang) to the end of its own last window -- the
      // window that holds its last gap cell -- or, for the last row of the plane, to its last own cell
      int lo = 0;
      if (i > 0) {
        const int php = (int)((dbase + (unsigned long long)((long long)(i - 1) * di)) & (unsigned long long)(U - 1));
        lo = ((di - 1 + php) / TJ + 1) * TJ - php - di;
      }
      const int hi = i == ei - 1 ? ej : ((di - 1 + ph) / TJ + 1) * TJ - ph;
      const int r = (U - 1) - ph + VW * c;  // LDS row of the lane's first element
      const int l = lb0 + r;
      E* q = dst + (long long)i * di + l;
      V out;
#pragma unroll
      for (int v = 0; v < VW; ++v) Lane<ES, VW>::set(out, v, tile[(r + v) * PITCH + ii]);
      if (l >= lo && l + VW <= hi) {
        storeVec<storePolicyOf<STREAM>(), ES * VW>(q, out);
      } else {
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (l + v >= lo && l + v < hi) storeVec<storePolicyOf<STREAM>(), ES>(q + v, Lane<ES, VW>::get(out, v));
      }
    }
  }
}

template <int STREAM>
void launchRowLinesT(int variant, int es, const Batch& b, unsigned int blocks, hipStream_t stream) {
  const dim3 grid(blocks), block(kThreads);
  if (es == 4) {
    if (variant == 4) transpose_rowlines_kernel<4, 4, 64, 128, STREAM, 128><<<grid, block, 0, stream>>>(b);
    else transpose_rowlines_kernel<4, 1, 64, 128, STREAM, 128><<<grid, block, 0, stream>>>(b);
  } else if (es == 8) {
    if (variant == 2) transpose_rowlines_kernel<8, 2, 64, 64, STREAM, 128><<<grid, block, 0, stream>>>(b);
    else transpose_rowlines_kernel<8, 1, 64, 64, STREAM, 128><<<grid, block, 0, stream>>>(b);
  } else {
    transpose_rowlines_kernel<16, 1, 32, 32, STREAM, 128><<<grid, block, 0, stream>>>(b);
  }
  CD_CHECK_HIP(hipGetLastError());
}

}  // namespace
}  // namespace kern

void launchRowLinesBatch(int es, int variant, int stream_access, const kern::Batch& b, unsigned int blocks, hipStream_t stream) {
  // local destinations only (the gap cells are read back): never the remote-store policy
  if (stream_access == 4 || stream_access == 2) kern::launchRowLinesT<4>(variant, es, b, blocks, stream);
  else kern::launchRowLinesT<0>(variant, es, b, blocks, stream);
}

}  // namespace cudecomp
