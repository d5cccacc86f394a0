// kernels_tile.h -- the LDS-tiled transposition of one TI x TJ tile and the kernel around it (device code; included by
// kernels_transpose.hip, which instantiates it per element size, and by the tuning harness scripts/tune/tune_fwd.hip).
//
// Replaces cutensorPermute, the closed-source 3-D permutation of the reference (include/internal/transpose.h:80-157):
//   transpose_kernel      fastest source dim != fastest destination dim.  A TI x TJ element tile is staged through LDS: global
//                         reads are coalesced along the source-fast dim, global writes along the destination-fast dim, both at
//                         16 B/lane when the tile edges hold whole vectors (VW elements per lane), element-wise otherwise.  The
//                         third dim is a batch index.  All 5 non-identity 3-D permutations with arbitrary (halo-padded,
//                         per-peer sub-block) strides reduce to this or to rows_kernel.
#pragma once
#include "kernels_dev.h"

namespace cudecomp {
namespace kern {

// LDS tile layout: row r (a source row, TI elements along i) is stored without padding; inside the row the
// VW-element groups (16 bytes for the vector variants) are permuted by XOR with the row's group index,
//   position(r, c) = r * TI + (((c / VW) ^ ((r / VW) % G)) * VW + c % VW),   G = TI / VW.
// Both phases then move whole 16-byte groups: the load phase writes the group it fetched, the store phase reads
// the VW x VW block (rows lj..lj+VW-1, one group) with VW vector reads, transposes it in registers and emits VW
// destination rows.  Lanes of a wavefront that work on the same group column sit in different rows and therefore,
// after the XOR, in different groups: every LDS access is a conflict-free 16-byte one (the previous padded
// layout spent half of its LDS cycles on bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5).
template <int ES, int VW, int TI, int TJ, int STREAM, bool GUARD>
__device__ __forceinline__ void transposeTile(Bytes<ES>* tile, const Bytes<ES>* __restrict__ src,
                                              Bytes<ES>* __restrict__ dst, long long i0, long long j0, long long ei,
                                              long long ej, long long sj, long long di, int tid) {
  using E = Bytes<ES>;
  using V = Bytes<ES * VW>;
  constexpr int G = TI / VW;            // groups per LDS row
  constexpr int TPR = TI / VW;          // lanes per source row segment
  constexpr int RPP = kThreads / TPR;   // source rows per pass
  constexpr int NP = TJ / RPP;          // load passes
  constexpr int TPO = TJ / VW;          // lanes per destination row segment
  constexpr int BPO = kThreads / TPO;   // VW-row blocks of destination rows per pass
  constexpr int NPO = TI / (BPO * VW);  // store passes
  V* vtile = reinterpret_cast<V*>(tile);
  // ---- global -> registers (all loads issued before the first use) -> LDS, rows along i
  {
    const int lg = tid % TPR;  // group index inside the row
    const int li = lg * VW;
    const int lj = tid / TPR;
    const E* base = src + (j0 + lj) * sj + i0 + li;
    V regs[NP] = {};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (!GUARD || (i0 + li < ei && j0 + lj + p * RPP < ej))
        regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(base + (long long)(p * RPP) * sj);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = lj + p * RPP;
      vtile[r * G + (lg ^ ((r / VW) % G))] = regs[p];
    }
  }
  __syncthreads();
  // ---- LDS -> registers (VW x VW block, transposed) -> global, rows along j
  {
    const int ljg = tid % TPO;  // group index along j: rows ljg*VW .. +VW-1 of the tile
    const int lj = ljg * VW;
    const int lb = tid / TPO;
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
      const int ig = lb + p * BPO;  // group along i: destination rows ig*VW .. +VW-1
      V in[VW];
#pragma unroll
      for (int v = 0; v < VW; ++v) in[v] = vtile[(lj + v) * G + (ig ^ (ljg % G))];
#pragma unroll
      for (int a = 0; a < VW; ++a) {
        V out;
#pragma unroll
        for (int v = 0; v < VW; ++v) Lane<ES, VW>::set(out, v, Lane<ES, VW>::get(in[v], a));
        const int ii = ig * VW + a;
        if (!GUARD || (i0 + ii < ei && j0 + lj < ej))
          storeVec<storePolicyOf<STREAM>(), ES * VW>(dst + (i0 + ii) * di + j0 + lj, out);
      }
    }
  }
}

// The padded layout (row pitch TI + 1 elements, element-wise LDS access): kept for 16-byte elements, where it is
// already conflict-free and measures faster than the swizzled one.
template <int ES, int VW, int TI, int TJ, int STREAM, bool GUARD>
__device__ __forceinline__ void transposeTilePadded(Bytes<ES>* tile, const Bytes<ES>* __restrict__ src,
                                              Bytes<ES>* __restrict__ dst, long long i0, long long j0, long long ei,
                                              long long ej, long long sj, long long di, int tid) {
  using E = Bytes<ES>;
  using V = Bytes<ES * VW>;
  constexpr int TPR = TI / VW;         // lanes per source row segment
  constexpr int RPP = kThreads / TPR;  // source rows per pass
  constexpr int NP = TJ / RPP;         // load passes
  constexpr int TPO = TJ / VW;         // lanes per destination row segment
  constexpr int RPO = kThreads / TPO;  // destination rows per pass
  constexpr int NPO = TI / RPO;        // store passes
  constexpr int PITCH = TI + 1;
  // ---- global -> registers (all loads issued before the first use) -> LDS, rows along i
  {
    const int li = (tid % TPR) * VW;
    const int lj = tid / TPR;
    const E* base = src + (j0 + lj) * sj + i0 + li;
    V regs[NP] = {};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (!GUARD || (i0 + li < ei && j0 + lj + p * RPP < ej))
        regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(base + (long long)(p * RPP) * sj);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      E* row = tile + (lj + p * RPP) * PITCH + li;
#pragma unroll
      for (int v = 0; v < VW; ++v) row[v] = Lane<ES, VW>::get(regs[p], v);
    }
  }
  __syncthreads();
  // ---- LDS -> registers -> global, rows along j
  {
    const int lj = (tid % TPO) * VW;
    const int li = tid / TPO;
    E* base = dst + (i0 + li) * di + j0 + lj;
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
      const int ii = li + p * RPO;
      V out;
#pragma unroll
      for (int v = 0; v < VW; ++v) Lane<ES, VW>::set(out, v, tile[(lj + v) * PITCH + ii]);
      if (!GUARD || (i0 + ii < ei && j0 + lj < ej))
        storeVec<storePolicyOf<STREAM>(), ES * VW>(base + (long long)(p * RPO) * di, out);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// transpose_kernel: dims (i, j, k): i is unit-stride in the source, j is unit-stride in the
// destination, k is the batch dim.  e = {ei, ej, ek}; ss = {1, sj, sk}; ds = {di, 1, dk} (elements).
// ---------------------------------------------------------------------------------------------
// STREAM: see storePolicyOf()
template <int ES, int VW, int TI, int TJ, int STREAM, bool SWZ>
__global__ __launch_bounds__(kThreads) void transpose_kernel(const Batch b) {
  using E = Bytes<ES>;
  static_assert(TI % VW == 0 && TJ % VW == 0, "tile must hold whole vectors");
  static_assert(kThreads % (TI / VW) == 0 && TJ % (kThreads / (TI / VW)) == 0, "load mapping");
  static_assert(kThreads % (TJ / VW) == 0 && TI % (kThreads / (TJ / VW) * VW) == 0, "store mapping");

  // XOR-swizzled without padding (transposeTile) or padded by one element per row (transposeTilePadded)
  __shared__ __attribute__((aligned(16))) E tile[SWZ ? TJ * TI : TJ * (TI + 1)];

  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const unsigned int ti_n = b.t0[mi], tj_n = b.t1[mi];
  // Workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only).  Give every XCD a
  // contiguous run of tiles, walked along i first: neighbouring tiles then extend the same source rows
  // inside ONE L2 / TLB domain instead of being dealt round-robin to all eight (measured on the 1024^3
  // fp64 permutations: 2.73 -> 2.66 ms strided-read side, 3.02 -> 2.93 ms strided-write side).
  const unsigned int nb = b.first_block[mi + 1] - b.first_block[mi];
  unsigned int lt = lb;
  if (b.p1[mi] & 1) {
    const unsigned int per = nb >> 3;
    if (lb < (per << 3)) lt = (lb & 7u) * per + (lb >> 3);
  }
  // Walk first along the tile dim that keeps the far-strided side on the same rows (same DRAM pages / TLB
  // entries): i first extends the source rows, j first extends the destination rows.
  unsigned int bi, bj, rest;
  if (b.p1[mi] & 2) {
    // j first, optionally in RUNS (kernels.cc classify(), "far-strided destination"): p0 = R > 1 and
    //   p1 bit 4 clear: R tiles along j, then all tile rows i, then the next R tiles along j;
    //   p1 bit 4 set:   all tiles along j, then R consecutive batch planes, then the tile rows i, then the next R planes.
    const unsigned int run = (unsigned int)b.p0[mi];
    if (run > 1 && !(b.p1[mi] & 4)) {
      const unsigned int jlo = lt % run;
      rest = lt / run;
      bi = rest % ti_n;
      rest /= ti_n;
      const unsigned int runs = tj_n / run;
      bj = (rest % runs) * run + jlo;
      rest /= runs;
    } else if (run > 1) {
      bj = lt % tj_n;
      rest = lt / tj_n;
      const unsigned int klo = rest % run;
      rest /= run;
      bi = rest % ti_n;
      rest = (rest / ti_n) * run + klo;
    } else {
      bj = lt % tj_n;
      rest = lt / tj_n;
      bi = rest % ti_n;
      rest /= ti_n;
    }
  } else {
    bi = lt % ti_n;
    rest = lt / ti_n;
    bj = rest % tj_n;
    rest /= tj_n;
  }
  const long long k = rest;

  const long long i0 = (long long)bi * TI, j0 = (long long)bj * TJ;
  const long long ei = m.e[0], ej = m.e[1];
  const long long sj = m.ss[1], sk = m.ss[2], di = m.ds[0], dk = m.ds[2];
  const E* __restrict__ src = reinterpret_cast<const E*>(m.src) + k * sk;
  E* __restrict__ dst = reinterpret_cast<E*>(m.dst) + k * dk;
  const int tid = threadIdx.x;

  // interior tiles skip every bounds test, which lets the compiler batch the 8 loads, the LDS traffic and
  // the 8 stores of a lane; edge tiles take the guarded copy of the same code
  if constexpr (SWZ) {
    if (i0 + TI <= ei && j0 + TJ <= ej) transposeTile<ES, VW, TI, TJ, STREAM, false>(tile, src, dst, i0, j0, ei, ej, sj, di, tid);
    else transposeTile<ES, VW, TI, TJ, STREAM, true>(tile, src, dst, i0, j0, ei, ej, sj, di, tid);
  } else {
    if (i0 + TI <= ei && j0 + TJ <= ej) transposeTilePadded<ES, VW, TI, TJ, STREAM, false>(tile, src, dst, i0, j0, ei, ej, sj, di, tid);
    else transposeTilePadded<ES, VW, TI, TJ, STREAM, true>(tile, src, dst, i0, j0, ei, ej, sj, di, tid);
  }
  if constexpr (STREAM == 3) remoteStoresDone();
}

}  // namespace kern
}  // namespace cudecomp
