// kernels_lines.hip -- the LDS-tiled transposition onto a halo-carrying pencil whose rows of CONSECUTIVE BATCH PLANES are
// adjacent in memory (the forward hops of an axis-contiguous cycle onto pencils with halos / padding): whole cache lines
// across the row ends.  One code object of its own (see kernels_batch.h for why there are several).
//
// Replaces, for this shape, cutensorPermute writing the output pencil of a single-rank transpose
// (reference include/internal/transpose.h:80-157, 326-362, 428-456).
#include "kernels_dev.h"

#include "errors.h"

namespace cudecomp {
namespace kern {
namespace {

// ---------------------------------------------------------------------------------------------
// transpose_lines_kernel.  Dims (i, j, k) as for transpose_kernel: i unit-stride in the source, j unit-stride in the
// destination, k the batch dim; e = {ei, ej, ek}; ss = {1, sj, sk}; ds = {di, 1, dk}.  Here dk is the destination pencil's
// ROW PITCH (row (i, k+1) follows row (i, k) in memory after g = dk - ej halo / padding cells) and di is far (one "slab" of
// ek rows per i).  transpose_window_kernel writes such rows in whole 64-byte units, but each ROW still begins and ends inside
// a 128-byte line, and a line that reaches HBM partly written costs about four line times (profiles/r05_tuning.md section 3:
// 0.60 of the HBM peak against 0.75 onto aligned rows).  The other part of those lines belongs to the next batch plane -- in
// the window kernel another workgroup, much later.
//
// Here j and k are fused into the destination's LINEAR position  l = k * dk + j,  0 <= l < L = (ek - 1) * dk + ej,  of slab i,
// and the tiles are windows of TJ linear positions: the window of slab i covers
//     l in [bl * TJ - p_i, bl * TJ - p_i + TJ),   p_i = element phase of the slab's first cell inside a UB-byte unit,
// so every store of the body is a whole aligned vector, a window is TJ * ES bytes of whole units ACROSS the row ends, and the
// cells of the gap between two rows (l mod dk >= ej: halo / padding cells of the output pencil, which the planner guarantees
// nobody else writes during the operation -- Move3D::dst_row_pitch, the contract of rows_dense_kernel) are read from the
// destination and written back unchanged.  Nothing is touched below a slab's first interior cell or above its last one (masked
// pieces there).  LDS row r of a tile holds linear position lb0 + r for the TI slabs of the tile: a source row (TI elements
// along i) when it is an interior cell, TI gathered destination cells when it is a gap cell.
// t0 = tiles along i, t1 = windows along l; p0 = run length of the tile walk (windows), p1 bit 1 = XCD-contiguous walk,
// p1 >> 8 = tile rows per group.
// ---------------------------------------------------------------------------------------------
template <int ES, int VW, int TI, int TJ, int STREAM, int UB>
__global__ __launch_bounds__(kThreads) void transpose_lines_kernel(const Batch b) {
  using E = Bytes<ES>;
  using V = Bytes<ES * VW>;
  constexpr int U = UB / ES;            // elements per alignment unit
  constexpr int ROWS = TJ + U - 1;      // linear positions a tile's windows can touch
  constexpr int PITCH = TI + 1;
  constexpr int TPR = TI / VW;          // lanes per source row segment
  constexpr int RPP = kThreads / TPR;   // LDS rows per load pass
  constexpr int NP = (ROWS + RPP - 1) / RPP;
  constexpr int TPO = TJ / VW;          // lanes per destination window
  constexpr int RPO = kThreads / TPO;   // slabs per store pass
  constexpr int NPO = TI / RPO;
  static_assert(kThreads % TPR == 0 && kThreads % TPO == 0 && TI % RPO == 0 && TJ % U == 0 && NP <= 64, "lines mapping");
  __shared__ __attribute__((aligned(16))) E tile[ROWS * PITCH];

  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const unsigned int ti_n = b.t0[mi], tl_n = b.t1[mi];
  const unsigned int nb = b.first_block[mi + 1] - b.first_block[mi];
  unsigned int lt = lb;
  if (b.p1[mi] & 1) {  // XCD-contiguous walk, see transpose_kernel
    const unsigned int per = nb >> 3;
    if (lb < (per << 3)) lt = (lb & 7u) * per + (lb >> 3);
  }
  // Walk, outermost to innermost: groups of G tile rows (p1 >> 8; 0 = all of them) -- runs of R windows along l (p0; 0 = the
  // whole range) -- the tile rows of the group -- the windows of the run.  The default (kernels.cc) is G = 16 with short runs
  // (2 KiB per slab; 32 KiB when the move has several groups): run after run, the 16 x TI slabs of the group each get their
  // next piece -- the source is then read in whole rows, plane by plane, and every slab's write stream advances steadily
  // (measured: profiles/r06_tuning.md).
  unsigned int bi, bl;
  {
    const unsigned int G = (unsigned int)(b.p1[mi] >> 8) ? (unsigned int)(b.p1[mi] >> 8) : ti_n;
    const unsigned int per_group = G * tl_n;
    const unsigned int g = lt / per_group, x = lt - g * per_group;
    const unsigned int gsize = (g + 1) * G <= ti_n ? G : ti_n - g * G;  // (the last group may have fewer rows)
    const unsigned int run = b.p0[mi] > 0 ? (unsigned int)b.p0[mi] : tl_n;
    const unsigned int full_runs = tl_n / run, full = full_runs * run * gsize;
    if (x < full) {
      const unsigned int lo = x % run, rest = x / run;
      bi = rest % gsize;
      bl = (rest / gsize) * run + lo;
    } else {
      const unsigned int tail = tl_n - full_runs * run, y = x - full;
      bl = full_runs * run + y % tail;
      bi = y / tail;
    }
    bi += g * G;
  }
  const int ei = (int)m.e[0], ej = (int)m.e[1];
  const int dk = (int)m.ds[2];
  const int L = (int)((m.e[2] - 1) * m.ds[2] + m.e[1]);
  const long long sj = m.ss[1], sk = m.ss[2], di = m.ds[0];
  const int i0 = (int)bi * TI, lb0 = (int)bl * TJ - (U - 1);  // LDS row 0 holds linear position lb0
  const E* __restrict__ src = reinterpret_cast<const E*>(m.src);
  E* dst = reinterpret_cast<E*>(m.dst);  // (read for the gap cells: no __restrict__)
  const int tid = threadIdx.x;
  const bool interior = i0 + TI <= ei && lb0 >= 0 && lb0 + ROWS <= L;

  // ---- global -> registers (all loads issued before the first use) -> LDS
  {
    const int li = (tid % TPR) * VW, lj = tid / TPR;
    int l = lb0 + lj, k = 0, j = l;
    if (l >= 0) {
      k = (int)((unsigned int)l / (unsigned int)dk);
      j = l - k * dk;
    }
    V regs[NP] = {};
    unsigned long long gap_passes = 0;  // passes in which this lane's LDS row is a gap row (NP is up to 40: element-wise lanes)
    if (interior) {
      // Interior tiles (nearly all): NO per-lane control flow around the loads -- a gap row loads the row's last interior
      // cell instead (a valid address; replaced below), so that the NP loads of a lane are all in flight before the first use.
      // (With the gap test around every load the compiler waited for each load before issuing the next: 15 % slower.)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        const bool gap = j >= ej;
        const int jc = gap ? ej - 1 : j;
        if (jj < ROWS) {
          regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(src + (long long)k * sk + (long long)jc * sj + i0 + li);
          if (gap) gap_passes |= 1ull << p;
        }
        j += RPP;
        if (j >= dk) {  // (dk >= ROWS: at most one row end per tile)
          j -= dk;
          ++k;
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int jj = lj + p * RPP;
        if (jj < ROWS && i0 + li < ei && l >= 0 && l < L) {
          if (j < ej) regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(src + (long long)k * sk + (long long)j * sj + i0 + li);
          else gap_passes |= 1ull << p;
        }
        l += RPP;
        j += RPP;
        if (j >= dk) {
          j -= dk;
          ++k;
        }
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
    // gap rows: a gap cell of every slab of the tile -- what the destination holds there goes back unchanged.  After the
    // source rows, by the lanes that own the LDS rows (program order: no barrier needed before they overwrite their own cells).
#ifdef CUDECOMP_TUNING_VARIANTS
    if (b.p1[mi] & 32) gap_passes = 0;  // (bit 32, measurements only: no gap gather -- the gap cells come back WRONG)
#endif
    if (gap_passes) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (gap_passes >> p & 1ull) {
          const int jj = lj + p * RPP;
          const int lg = lb0 + jj;
          E* row = tile + jj * PITCH + li;
#pragma unroll
          for (int v = 0; v < VW; ++v)
            if (interior || i0 + li + v < ei) row[v] = loadVec<false, ES>(dst + (long long)(i0 + li + v) * di + lg);
        }
      }
    }
  }
  __syncthreads();
  // ---- LDS -> registers -> global: slab i takes LDS rows (U-1) - p_i ... + TJ
  {
    const int c = tid % TPO, lr = tid / TPO;
    const unsigned long long dbase = (unsigned long long)(reinterpret_cast<uintptr_t>(dst)) / ES;
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
      const int ii = lr + p * RPO;
      const int i = i0 + ii;
      if (!interior && i >= ei) continue;
      const int ph = (int)((dbase + (unsigned long long)((long long)i * di)) & (unsigned long long)(U - 1));
      const int r = (U - 1) - ph + VW * c;  // LDS row of the lane's first element
      const int l = lb0 + r;
      E* q = dst + (long long)i * di + l;
      V out;
#pragma unroll
      for (int v = 0; v < VW; ++v) Lane<ES, VW>::set(out, v, tile[(r + v) * PITCH + ii]);
      if (interior || (l >= 0 && l + VW <= L)) {
        storeVec<storePolicyOf<STREAM>(), ES * VW>(q, out);
      } else {
#pragma unroll
        for (int v = 0; v < VW; ++v)
          if (l + v >= 0 && l + v < L) storeVec<storePolicyOf<STREAM>(), ES>(q + v, Lane<ES, VW>::get(out, v));
      }
    }
  }
}

template <int STREAM, int UB>
void launchLinesT(int variant, int es, const Batch& b, unsigned int blocks, hipStream_t stream) {
  const dim3 grid(blocks), block(kThreads);
  if (es == 4) {
    if (variant == 4) transpose_lines_kernel<4, 4, 64, 128, STREAM, UB><<<grid, block, 0, stream>>>(b);
    else transpose_lines_kernel<4, 1, 64, 128, STREAM, UB><<<grid, block, 0, stream>>>(b);
  } else if (es == 8) {
    if (variant == 2) transpose_lines_kernel<8, 2, 64, 64, STREAM, UB><<<grid, block, 0, stream>>>(b);
    else transpose_lines_kernel<8, 1, 64, 64, STREAM, UB><<<grid, block, 0, stream>>>(b);
  } else {
    transpose_lines_kernel<16, 1, 32, 32, STREAM, UB><<<grid, block, 0, stream>>>(b);
  }
  CD_CHECK_HIP(hipGetLastError());
}

}  // namespace
}  // namespace kern

int linesUnitBytes(int unit_choice) {
#ifdef CUDECOMP_TUNING_VARIANTS
  if (unit_choice == 64) return 64;
#endif
  (void)unit_choice;
  return 128;
}

void launchLinesBatch(int es, int variant, int stream_access, int unit_bytes, const kern::Batch& b, unsigned int blocks,
                      hipStream_t stream) {
  // local destinations only (the gap cells are read back): never the remote-store policy
  const bool streaming = stream_access == 4 || stream_access == 2;
#ifdef CUDECOMP_TUNING_VARIANTS
  if (unit_bytes == 64) {
    if (streaming) kern::launchLinesT<4, 64>(variant, es, b, blocks, stream);
    else kern::launchLinesT<0, 64>(variant, es, b, blocks, stream);
    return;
  }
#endif
  (void)unit_bytes;
  if (streaming) kern::launchLinesT<4, 128>(variant, es, b, blocks, stream);
  else kern::launchLinesT<0, 128>(variant, es, b, blocks, stream);
}

}  // namespace cudecomp
