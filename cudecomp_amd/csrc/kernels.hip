// kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) data-movement kernels.
//
// These replace the two device code paths of NVIDIA/cuDecomp's local phases:
//   * the batched strided 3-D copy kernel (reference include/internal/cudecomp_kernels.cuh:125-180:
//     one element per thread per iteration, two 64-bit div/mod pairs per element, no vector access);
//   * cutensorPermute, the closed-source 3-D permutation (reference include/internal/transpose.h:80-157).
// Here both are one object, a Move3D (plan.h), executed by one of three kernels:
//
//   rows_kernel<VB>       fastest dim contiguous on both sides.  Each lane moves VB = 16 (8, 4) bytes,
//                         a 256-thread workgroup keeps 4 vectors per lane (16 KiB) in flight, lanes run
//                         along the row so that every wavefront touches 1 KiB contiguous segments.
//                         No per-element index math: one (row, plane) decode per WORKGROUP.
//   transpose_kernel      fastest source dim != fastest destination dim.  A TI x TJ element tile is
//                         staged through LDS: global reads are coalesced along the source-fast dim,
//                         global writes along the destination-fast dim, both at 16 B/lane when the
//                         addresses allow (VW elements per lane), element-wise otherwise.  The third
//                         dim is a batch index.  All 5 non-identity 3-D permutations with arbitrary
//                         (halo-padded, per-peer sub-block) strides reduce to this or to rows_kernel.
//   generic_kernel<ES>    degenerate shapes (no unit stride on one side, 1-element rows).
//
// This is pure data movement: no MFMA; the bound is HBM (8 TB/s spec, ~6.3 TB/s achievable copy rate).
// Up to kMaxBatch moves (e.g. the per-peer pack copies of one transpose) share one launch; the
// descriptors travel in the kernel argument segment.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "errors.h"
#include "kernels.h"

namespace cudecomp {

namespace {

constexpr int kMaxBatch = 8;
constexpr int kThreads = 256;
constexpr int kRowsUnroll = 4;
constexpr long long kStreamBytes = 32ll << 20;  // moves at least this large use non-temporal access

struct DevMove {
  const char* src;
  char* dst;
  long long e[3];   // extents   (units depend on the kernel, see the launchers)
  long long ss[3];  // src strides
  long long ds[3];  // dst strides
};

struct Batch {
  int n;
  int interleave;  // 1: workgroup b serves move b % n (moves with REMOTE destinations: keeps every xGMI link busy
                   // for the whole launch instead of draining one peer's chunk after the other)
  int p0[kMaxBatch];                      // kernel-specific small parameter
  int p1[kMaxBatch];                      // second small parameter (transpose: XCD-contiguous tile walk)
  unsigned int first_block[kMaxBatch + 1];
  unsigned int t0[kMaxBatch];             // tiles along dim 0
  unsigned int t1[kMaxBatch];             // tiles along dim 1
  DevMove m[kMaxBatch];
};

// N-byte lane payloads as native vector types (kept in VGPRs; a struct-of-array payload gets
// "promoted" to LDS by the compiler, which costs occupancy and LDS bandwidth).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int N> struct BytesOf;
template <> struct BytesOf<4> { using type = unsigned int; };
template <> struct BytesOf<8> { using type = u32x2; };
template <> struct BytesOf<16> { using type = u32x4; };
template <int N> using Bytes = typename BytesOf<N>::type;

// The same payloads for GLOBAL memory at element (not vector) alignment.  gfx950 global loads/stores of 8 and
// 16 bytes only need dword-aligned addresses, so a halo-shifted pencil (interior starting one fp64 past a 16-byte
// boundary, say) still moves 16 bytes per lane; a wavefront then touches one extra cache line per KiB.
typedef u32x2 __attribute__((aligned(4))) u32x2_g;
typedef u32x4 __attribute__((aligned(4))) u32x4_g;
template <int N> struct GlobalBytesOf;
template <> struct GlobalBytesOf<4> { using type = unsigned int; };
template <> struct GlobalBytesOf<8> { using type = u32x2_g; };
template <> struct GlobalBytesOf<16> { using type = u32x4_g; };
template <int N> using GlobalBytes = typename GlobalBytesOf<N>::type;

// element v (ES bytes) of a VW-element vector
template <int ES, int VW> struct Lane;
template <int ES> struct Lane<ES, 1> {
  static __device__ __forceinline__ Bytes<ES> get(const Bytes<ES>& x, int) { return x; }
  static __device__ __forceinline__ void set(Bytes<ES>& x, int, const Bytes<ES>& e) { x = e; }
};
template <> struct Lane<4, 4> {
  static __device__ __forceinline__ unsigned int get(const u32x4& x, int v) { return x[v]; }
  static __device__ __forceinline__ void set(u32x4& x, int v, unsigned int e) { x[v] = e; }
};
template <> struct Lane<4, 2> {
  static __device__ __forceinline__ unsigned int get(const u32x2& x, int v) { return x[v]; }
  static __device__ __forceinline__ void set(u32x2& x, int v, unsigned int e) { x[v] = e; }
};
template <> struct Lane<8, 2> {
  static __device__ __forceinline__ u32x2 get(const u32x4& x, int v) { return v == 0 ? x.xy : x.zw; }
  static __device__ __forceinline__ void set(u32x4& x, int v, const u32x2& e) {
    if (v == 0) x.xy = e; else x.zw = e;
  }
};

// Streaming (non-temporal) access for moves far larger than the caches: measured +3..15 % on the 1024^3
// permutations (profiles/r01_tuning.md); small moves keep the default policy so a following kernel can
// still find the data in L2 / Infinity Cache.
template <bool STREAM, int N>
__device__ __forceinline__ Bytes<N> loadVec(const void* p) {
  const GlobalBytes<N>* q = static_cast<const GlobalBytes<N>*>(p);
  if constexpr (STREAM) return __builtin_nontemporal_load(q);
  else return *q;
}
// Store policies: ST_CACHED default, ST_STREAM non-temporal, ST_REMOTE system-scope write-through (sc0 sc1) for
// destinations in ANOTHER GPU's memory (one-sided puts over xGMI).  A plain store to peer memory may linger as a
// dirty line in this XCD's L2 until some later system-scope release; the stream-ordered exchanges signal the
// receiver from the NEXT kernel on the stream, whose release would only write back the L2 of the one XCD it runs
// on.  Write-through stores need no flush: once the wave's stores are acknowledged (s_waitcnt vmcnt(0) at the end of
// the kernel, remoteStoresDone()) they are in the peer's memory.  (A `volatile` store gives the same cache bits but
// makes the compiler wait for every single store, which serialises a lane's 4-8 stores.)
enum StorePolicy { ST_CACHED = 0, ST_STREAM = 1, ST_REMOTE = 2 };
template <int N> __device__ __forceinline__ void storeRemote(void* p, const Bytes<N>& v);
template <> __device__ __forceinline__ void storeRemote<4>(void* p, const Bytes<4>& v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void storeRemote<8>(void* p, const Bytes<8>& v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <> __device__ __forceinline__ void storeRemote<16>(void* p, const Bytes<16>& v) {
  // gfx940+ hazard: a VALU write to the data VGPRs of a store wider than 64 bits needs 2 wait states after the
  // store.  The compiler inserts them for its own stores but cannot see inside inline assembly (without the s_nop a
  // few cells per GiB arrived holding the next tile's address arithmetic instead of data).
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void remoteStoresDone() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int POLICY, int N>
__device__ __forceinline__ void storeVec(void* p, const Bytes<N>& v) {
  GlobalBytes<N>* q = static_cast<GlobalBytes<N>*>(p);
  if constexpr (POLICY == ST_REMOTE) storeRemote<N>(p, v);
  else if constexpr (POLICY == ST_STREAM) __builtin_nontemporal_store(v, q);
  else *q = v;
}
// STREAM template parameter of the kernels: 0 = default caching, 1 = non-temporal loads, 2 = non-temporal loads
// and stores, 3 = non-temporal loads + system-scope write-through stores (remote destinations), 4 = cached loads +
// non-temporal stores (misaligned sources: neighbouring tiles share the partially used lines through L2)
template <int STREAM> constexpr int storePolicyOf() {
  return STREAM == 3 ? ST_REMOTE : ((STREAM == 2 || STREAM == 4) ? ST_STREAM : ST_CACHED);
}
template <int STREAM> constexpr bool loadsStream() { return STREAM >= 1 && STREAM <= 3; }

__device__ __forceinline__ int findMove(const Batch& b, unsigned int block) {
  int mi = 0;
#pragma unroll
  for (int i = 1; i < kMaxBatch; ++i)
    if (i < b.n && block >= b.first_block[i]) mi = i;
  return mi;
}

// workgroup -> (move, workgroup index inside the move); false for the filler workgroups of an interleaved launch
__device__ __forceinline__ bool locate(const Batch& b, unsigned int block, int& mi, unsigned int& lb) {
  if (b.interleave) {
    mi = (int)(block % (unsigned int)b.n);
    lb = block / (unsigned int)b.n;
    return lb < b.first_block[mi + 1] - b.first_block[mi];
  }
  mi = findMove(b, block);
  lb = block - b.first_block[mi];
  return true;
}

// ---------------------------------------------------------------------------------------------
// rows_kernel: e[0] = vectors per row, e[1] = rows, e[2] = planes; ss/ds[1], [2] in BYTES.
// p0 = log2(lanes per row).  A workgroup covers (256 >> p0) * kRowsUnroll rows x (1 << p0) vectors.
// ---------------------------------------------------------------------------------------------
template <int VB, int STREAM>
__global__ __launch_bounds__(kThreads) void rows_kernel(const Batch b) {
  using V = Bytes<VB>;
  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const int lg = b.p0[mi];
  const int lpr = 1 << lg;
  const int rb = kThreads >> lg;
  const unsigned int tc = b.t0[mi], tr = b.t1[mi];
  const unsigned int bc = lb % tc;
  const unsigned int rest = lb / tc;
  const unsigned int br = rest % tr;
  const long long plane = rest / tr;

  const long long col = (long long)bc * lpr + (threadIdx.x & (lpr - 1));
  const long long r0 = (long long)br * rb * kRowsUnroll + (threadIdx.x >> lg);
  if (col >= m.e[0]) return;
  const char* __restrict__ s = m.src + plane * m.ss[2] + col * VB;
  char* __restrict__ d = m.dst + plane * m.ds[2] + col * VB;

  V v[kRowsUnroll];
#pragma unroll
  for (int u = 0; u < kRowsUnroll; ++u) {
    const long long r = r0 + (long long)u * rb;
    if (r < m.e[1]) v[u] = loadVec<(STREAM >= 1), VB>(s + r * m.ss[1]);
  }
#pragma unroll
  for (int u = 0; u < kRowsUnroll; ++u) {
    const long long r = r0 + (long long)u * rb;
    if (r < m.e[1]) storeVec<(STREAM == 3 ? ST_REMOTE : (STREAM >= 1 ? ST_STREAM : ST_CACHED)), VB>(d + r * m.ds[1], v[u]);
  }
  if constexpr (STREAM == 3) remoteStoresDone();
}

// ---------------------------------------------------------------------------------------------
// rows_shifted_kernel: the same copy for DESTINATION rows that do not start on 64-byte boundaries (unpacks into halo-
// carrying pencils, halo faces).  With the plain lane mapping every wavefront's 1-KiB run begins and ends inside a
// 64-byte unit of the destination, and the two store instructions that share a unit each write part of it.  Here the
// lanes of a row are laid out from the 64-byte boundary BELOW the row's start: lane `col` covers destination bytes
// [col*VB - shift, +VB) of the row, shift = (row address mod 64) -- every full vector is aligned and whole units are
// written by one instruction; only the two ends of each ROW are partial (copied in 4-byte pieces).  Loads take the
// misalignment instead, which costs nothing measurable (profiles/r02_tuning.md: 8 GiB onto halo-shifted rows
// 3.3-3.4 ms -> 3.0 ms in the probe).  e[0] = vectors per row INCLUDING one unit of slack, e[1] = rows, e[2] = planes;
// p1 = row length in bytes.
// ---------------------------------------------------------------------------------------------
template <int VB> __device__ __forceinline__ unsigned int getDword(const Bytes<VB>& x, int k) {
  if constexpr (VB == 4) return x;
  else return x[k];
}
template <int VB> __device__ __forceinline__ void setDword(Bytes<VB>& x, int k, unsigned int e) {
  if constexpr (VB == 4) x = e;
  else x[k] = e;
}

template <int VB, int STREAM>
__global__ __launch_bounds__(kThreads) void rows_shifted_kernel(const Batch b) {
  using V = Bytes<VB>;
  constexpr int POLICY = STREAM == 3 ? ST_REMOTE : (STREAM >= 1 ? ST_STREAM : ST_CACHED);
  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const int lg = b.p0[mi];
  const int lpr = 1 << lg;
  const int rb = kThreads >> lg;
  const unsigned int tc = b.t0[mi], tr = b.t1[mi];
  const unsigned int bc = lb % tc;
  const unsigned int rest = lb / tc;
  const unsigned int br = rest % tr;
  const long long plane = rest / tr;
  const long long row_bytes = b.p1[mi];
  // e[0] = vectors of a row + one 64-byte unit of slack (the shift moves up to a unit's worth past the row's own
  // vectors).  When the slack needs a tile column of its own that column is almost empty; letting the first lanes of
  // the last full column take it in a second step instead was measured and is far worse (3.0 -> 4.2 ms on 8 GiB: those
  // workgroups pay two memory round trips).
  const long long col = (long long)bc * lpr + (threadIdx.x & (lpr - 1));
  const long long r0 = (long long)br * rb * kRowsUnroll + (threadIdx.x >> lg);
  if (col >= m.e[0]) return;
  const char* __restrict__ s = m.src + plane * m.ss[2];
  char* __restrict__ d = m.dst + plane * m.ds[2];

  V v[kRowsUnroll] = {};
  long long off[kRowsUnroll];
#pragma unroll
  for (int u = 0; u < kRowsUnroll; ++u) {
    const long long r = r0 + (long long)u * rb;
    off[u] = -2 * VB;  // "nothing to do"
    if (r < m.e[1]) {
      const long long shift = (long long)(reinterpret_cast<uintptr_t>(d + r * m.ds[1]) & 63);
      off[u] = col * VB - shift;
      const char* sr = s + r * m.ss[1] + off[u];
      if (off[u] >= 0 && off[u] + VB <= row_bytes) {
        v[u] = loadVec<(STREAM >= 1), VB>(sr);
      } else {  // a row end: only the 4-byte pieces of my vector that lie inside the row (all loads in this phase)
#pragma unroll
        for (int k = 0; k < VB / 4; ++k)
          if (off[u] + 4 * k >= 0 && off[u] + 4 * k < row_bytes) setDword<VB>(v[u], k, *reinterpret_cast<const unsigned int*>(sr + 4 * k));
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kRowsUnroll; ++u) {
    const long long r = r0 + (long long)u * rb;
    if (r >= m.e[1]) continue;
    char* dr = d + r * m.ds[1] + off[u];
    if (off[u] >= 0 && off[u] + VB <= row_bytes) {
      storeVec<POLICY, VB>(dr, v[u]);
    } else {
#pragma unroll
      for (int k = 0; k < VB / 4; ++k)
        if (off[u] + 4 * k >= 0 && off[u] + 4 * k < row_bytes) storeVec<POLICY, 4>(dr + 4 * k, getDword<VB>(v[u], k));
    }
  }
  if constexpr (STREAM == 3) remoteStoresDone();
}

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
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int jj = lj + p * RPP;
      const long long j = jb + jj;
      if (jj < ROWS && (interior || (i0 + li < ei && j >= 0 && j < ej)))
        regs[p] = loadVec<loadsStream<STREAM>(), ES * VW>(base + (long long)(p * RPP) * sj);
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

// ---------------------------------------------------------------------------------------------
// generic_kernel: element-wise, lanes along dim p0 (the destination-fast dim when there is one).
// ---------------------------------------------------------------------------------------------
template <int ES, bool REMOTE>
__global__ __launch_bounds__(kThreads) void generic_kernel(const Batch b) {
  using E = Bytes<ES>;
  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const unsigned int nb = b.first_block[mi + 1] - b.first_block[mi];
  const int f = b.p0[mi], g = (f + 1) % 3, h = (f + 2) % 3;
  const unsigned long long ef = m.e[f], eg = m.e[g];
  const unsigned long long total = ef * eg * (unsigned long long)m.e[h];
  const E* __restrict__ src = reinterpret_cast<const E*>(m.src);
  E* __restrict__ dst = reinterpret_cast<E*>(m.dst);
  for (unsigned long long n = (unsigned long long)lb * kThreads + threadIdx.x; n < total;
       n += (unsigned long long)nb * kThreads) {
    const unsigned long long kf = n % ef, t = n / ef;
    const unsigned long long kg = t % eg, kh = t / eg;
    storeVec<(REMOTE ? ST_REMOTE : ST_CACHED), ES>(dst + (kf * m.ds[f] + kg * m.ds[g] + kh * m.ds[h]),
                                                     src[kf * m.ss[f] + kg * m.ss[g] + kh * m.ss[h]]);
  }
  if constexpr (REMOTE) remoteStoresDone();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Classified {
  MoveClass cls;
  int variant;  // rows: vector bytes; transpose: elements per vector
  DevMove dm;
  int p0, p1;
  int stream;  // 0 default caching, 1 streaming loads, 2 streaming loads + stores, 3 streaming loads + remote stores
  bool swizzle = false;  // transposes: XOR-swizzled LDS tile (else padded rows)
  bool window = false;   // transposes: destination rows off the 64-byte grid -> transpose_window_kernel
  unsigned int t0, t1;
  unsigned long long blocks;
  i64 elements;
};

int ilog2ceil(long long x) {
  int l = 0;
  while ((1LL << l) < x) ++l;
  return l;
}

template <int ES>
constexpr int tileI() {
  return ES == 16 ? 32 : 64;
}

Classified classify(const Move3D& in, void* const bufs[3], int es, const KernelTuning* tuning, void* dst_base,
                    bool remote) {
  Move3D m = in;
  normalizeMove(m);
  Classified c{};
  c.elements = m.elements();
  c.stream = ((c.elements * es >= kStreamBytes || (tuning && tuning->force_streaming)) && !(tuning && tuning->no_streaming)) ? 2 : 0;
  if (remote) c.stream = 3;  // destination in a peer's memory: write-through stores, whatever the size
  // diagnostic override of the store policy of local moves (the 8-shared-rank hunt)
  const int local_policy = (tuning && !remote) ? tuning->local_store_policy : -1;
  if (local_policy == 0) c.stream = 0;
  else if (local_policy == 1) c.stream = 2;
  else if (local_policy == 2) c.stream = 3;
  c.dm.src = static_cast<const char*>(bufs[m.src_buf]) + m.src_off * es;
  c.dm.dst = static_cast<char*>(dst_base ? dst_base : bufs[m.dst_buf]) + m.dst_off * es;
  const bool force_generic = tuning && tuning->force_class == MOVE_GENERIC;

  if (!force_generic && m.ss[0] <= 1 && m.ds[0] <= 1) {
    // rows contiguous on both sides (also the all-extents-1 case).  Widest vector that divides the row length;
    // addresses only need the element's natural alignment (see GlobalBytes).
    int vb = 16;
    while (vb > es && (m.extent[0] * es) % vb != 0) vb >>= 1;
    c.cls = MOVE_ROWS_VEC;
    c.variant = vb;
    c.dm.e[0] = m.extent[0] * es / vb;
    c.dm.e[1] = m.extent[1];
    c.dm.e[2] = m.extent[2];
    for (int i = 1; i < 3; ++i) {
      c.dm.ss[i] = m.ss[i] * es;
      c.dm.ds[i] = m.ds[i] * es;
    }
    // rows that land off the 64-byte grid (and are long enough for it to matter): lanes laid out from the unit boundary
    // below each row's start (rows_shifted_kernel), one unit of slack vectors per row
    const uintptr_t dst_bits = reinterpret_cast<uintptr_t>(c.dm.dst) | (uintptr_t)c.dm.ds[1] | (uintptr_t)c.dm.ds[2];
    const int shift_mode = tuning ? tuning->window_mode : -1;
    if ((dst_bits & 63) != 0 && m.extent[0] * es >= 256 && shift_mode != 0 && (shift_mode == 1 || c.elements * es >= (1ll << 20))) {
      c.window = true;
      c.p1 = (int)(m.extent[0] * es);  // row length in bytes (rows longer than 2 GiB keep the plain kernel)
      if (m.extent[0] * es > 0x7fffffffLL) c.window = false;
    }
    if (c.window) c.dm.e[0] += 64 / vb;  // one unit of slack vectors per row (see rows_shifted_kernel)
    c.p0 = std::min(8, ilog2ceil(c.dm.e[0]));
    const long long lpr = 1LL << c.p0, rows_per_block = (long long)(kThreads >> c.p0) * kRowsUnroll;
    c.t0 = (unsigned int)((c.dm.e[0] + lpr - 1) / lpr);
    c.t1 = (unsigned int)((c.dm.e[1] + rows_per_block - 1) / rows_per_block);
    c.blocks = (unsigned long long)c.t0 * c.t1 * (unsigned long long)c.dm.e[2];
    return c;
  }

  int t = -1;
  if (!force_generic && m.ss[0] == 1) {
    if (m.ds[1] == 1) t = 1;
    if (m.ds[2] == 1) t = 2;
  }
  if (t > 0 && m.extent[0] >= 4 && m.extent[t] >= 4) {
    const int k = 3 - t;
    c.cls = MOVE_TRANSPOSE;
    c.dm.e[0] = m.extent[0];
    c.dm.e[1] = m.extent[t];
    c.dm.e[2] = m.extent[k];
    c.dm.ss[0] = 1;
    c.dm.ss[1] = m.ss[t];
    c.dm.ss[2] = m.ss[k];
    c.dm.ds[0] = m.ds[0];
    c.dm.ds[1] = 1;
    c.dm.ds[2] = m.ds[k];
    // 16 bytes per lane whenever both tile edges hold whole vectors (no alignment requirement, see GlobalBytes)
    int vw = 16 / es;
    if (c.dm.e[0] % vw != 0 || c.dm.e[1] % vw != 0) vw = 1;
    c.variant = vw;
    c.p1 = (tuning && tuning->xcd_walk == 0) ? 0 : 1;  // XCD-contiguous tile walk
    c.swizzle = (tuning && tuning->lds_swizzle >= 0) ? tuning->lds_swizzle != 0 : es != 16;
    // Tile walk order inside an XCD's run: j first makes consecutive tiles extend the same DESTINATION rows
    // (contiguous write stream per row), i first the same source rows.  Measured on 8 GiB permutations
    // (profiles/r01_tuning.md): j first wins or ties for line-aligned moves (8-11 % at 16-byte elements and on
    // the strided-read side at 4-byte elements; 4-byte moves whose destination rows are the far-strided side
    // lose 1-3 % and keep i first), i first wins by 5-10 % for misaligned moves, where L2 merges the
    // partially read lines of neighbouring tiles.
    bool j_first = es != 4 || c.dm.ss[1] > c.dm.ds[0];
    // Rows that do not start on cache-line boundaries (halo-shifted or odd-extent pencils) leave partially covered
    // lines at both ends of every tile row.
    //  * Misaligned SOURCE rows only: the partially used lines are shared with the neighbouring tile; cached loads let
    //    L2 serve the second use (non-temporal loads fetch them twice), the aligned stores keep streaming.
    //  * Misaligned DESTINATION rows: partial 64-byte units written by two tiles are what costs (a cached store lets L2
    //    merge some: fp32 3.0 -> 4.4 TB/s, fp64 3.9 -> 4.8 TB/s on a halo-shifted 8 GiB permutation); the window kernel
    //    writes whole units instead (4.8 -> 5.1-5.3 TB/s), with streaming stores.
    const uintptr_t src_bits = reinterpret_cast<uintptr_t>(c.dm.src) | (uintptr_t)(c.dm.ss[1] * es) | (uintptr_t)(c.dm.ss[2] * es);
    const uintptr_t dst_bits = reinterpret_cast<uintptr_t>(c.dm.dst) | (uintptr_t)(c.dm.ds[0] * es) | (uintptr_t)(c.dm.ds[2] * es);
    const uintptr_t align_req = (tuning && tuning->stream_alignment > 0) ? (uintptr_t)tuning->stream_alignment : 128;
    const bool src_mis = src_bits % align_req != 0, dst_mis = dst_bits % 64 != 0;
    const int window_mode = tuning ? tuning->window_mode : -1;  // -1 auto, 0 never, 1 whenever the destination is misaligned
    c.window = dst_mis && window_mode != 0 && (window_mode == 1 || c.elements * es >= (1ll << 20));
    if (c.window) {
      if (c.stream == 2) c.stream = 4;  // cached loads (the overlap rows hit in L2), streaming whole-unit stores
      j_first = true;
    } else if (src_mis || dst_bits % align_req != 0) {
      if (c.stream == 2) {
        if (dst_bits % align_req != 0) c.stream = (tuning && tuning->misaligned_store_mode >= 0) ? tuning->misaligned_store_mode : 0;
        else c.stream = 4;
      }
      j_first = false;
    }
    // One measured outlier: 16-byte elements whose destination batch stride is not a multiple of 4 KiB (rows padded by a
    // cache line) lose a third of their rate with streaming stores (8 GiB permutation: 4.0 ms streaming, 3.4 ms cached;
    // 4- and 8-byte elements with the same padding prefer streaming, profiles/r02_tuning.md).
    if (es == 16 && c.stream == 2 && !c.window && c.dm.e[2] > 1 && ((uintptr_t)(c.dm.ds[2] * es) % 4096) != 0) c.stream = 0;
    if (tuning && tuning->stream_mode >= 0 && c.stream != 3 && c.stream != 0) c.stream = tuning->stream_mode;
    if (local_policy == 0) c.stream = 0;
    else if (local_policy == 1 && c.stream == 0) c.stream = 2;
    if (tuning && tuning->walk_order >= 0) j_first = tuning->walk_order == 1;
    if (j_first) c.p1 |= 2;
    // (window kernel, 4-byte elements: 64 x 128 tiles -- a 64-byte unit is 16 elements, the longer window halves the
    // share of overlap rows)
    // (window kernel, 8-byte elements, optional: 128 x 64 tiles with 512 threads -- 1-KiB source segments span nine
    // lines instead of 2 x five; variant 102)
#ifdef CUDECOMP_TUNING_VARIANTS
    const bool wide = c.window && es == 8 && vw == 2 && tuning && tuning->window_wide == 1;
#else
    const bool wide = false;  // (the 128 x 64 / 512-thread window variant exists in tuning builds only)
#endif
    if (wide) c.variant = 102;
    // (4-byte elements, 16-byte lanes, plain kernel: optional 128 x 64 / 64 x 128 tiles -- variants 204 / 304)
    int shape = 0;
    if (!c.window && es == 4 && vw == 4) shape = 2;
#ifdef CUDECOMP_TUNING_VARIANTS
    if (!c.window && es == 4 && vw == 4 && tuning && tuning->tile_shape >= 0) shape = tuning->tile_shape;
#endif
    if (shape == 1) c.variant = 204;
    else if (shape == 2) c.variant = 304;
    // (shape 0 keeps variant 4 = 64 x 64 tiles: only in builds with CUDECOMP_TUNING_VARIANTS)
    const int ti = (es == 16) ? 32 : ((wide || shape == 1) ? 128 : 64);
    const int tj = ((c.window && es == 4) || shape == 2) ? 128 : ((es == 16) ? 32 : 64);
    c.t0 = (unsigned int)((c.dm.e[0] + ti - 1) / ti);
    c.t1 = (unsigned int)((c.dm.e[1] + (c.window ? 64 / es - 1 : 0) + tj - 1) / tj);
    c.blocks = (unsigned long long)c.t0 * c.t1 * (unsigned long long)c.dm.e[2];
    return c;
  }

  c.cls = MOVE_GENERIC;
  c.variant = es;
  for (int i = 0; i < 3; ++i) {
    c.dm.e[i] = m.extent[i];
    c.dm.ss[i] = m.ss[i];
    c.dm.ds[i] = m.ds[i];
  }
  c.p0 = 0;
  for (int i = 0; i < 3; ++i)
    if (m.ds[i] == 1 && m.extent[i] > 1) c.p0 = i;
  const unsigned long long want = ((unsigned long long)c.elements + kThreads - 1) / kThreads;
  c.blocks = std::min<unsigned long long>(std::max<unsigned long long>(want, 1), 8192);
  return c;
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
  CD_CHECK_HIP(hipGetLastError());
}

template <int STREAM, bool SWZ>
void launchBatchT(MoveClass cls, int variant, int es, const Batch& b, unsigned int blocks, hipStream_t stream, bool window = false) {
  const dim3 grid(blocks), block(kThreads);
  constexpr int ROWS_STREAM = STREAM == 3 ? 3 : (STREAM >= 1 ? 1 : 0);  // (4 only occurs for transposes)
  switch (cls) {
    case MOVE_ROWS_VEC:
      if (window) {
        if (variant == 16) rows_shifted_kernel<16, ROWS_STREAM><<<grid, block, 0, stream>>>(b);
        else if (variant == 8) rows_shifted_kernel<8, ROWS_STREAM><<<grid, block, 0, stream>>>(b);
        else rows_shifted_kernel<4, ROWS_STREAM><<<grid, block, 0, stream>>>(b);
      } else if (variant == 16) rows_kernel<16, ROWS_STREAM><<<grid, block, 0, stream>>>(b);
      else if (variant == 8) rows_kernel<8, ROWS_STREAM><<<grid, block, 0, stream>>>(b);
      else rows_kernel<4, ROWS_STREAM><<<grid, block, 0, stream>>>(b);
      break;
    case MOVE_TRANSPOSE:
      if (es == 4) {
#ifdef CUDECOMP_TUNING_VARIANTS  // tuning builds only (make TUNING_VARIANTS=1): the 64 x 64 and 128 x 64 tiles of the A/B
        if (variant == 204) transpose_kernel<4, 4, 128, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
        else if (variant == 4) transpose_kernel<4, 4, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
        else
#endif
        if (variant == 304) transpose_kernel<4, 4, 64, 128, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
        else transpose_kernel<4, 1, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
      } else if (es == 8) {
        if (variant == 2) transpose_kernel<8, 2, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
        else transpose_kernel<8, 1, 64, 64, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
      } else {
        transpose_kernel<16, 1, 32, 32, STREAM, SWZ><<<grid, block, 0, stream>>>(b);
      }
      break;
    default:
      if (es == 4) generic_kernel<4, STREAM == 3><<<grid, block, 0, stream>>>(b);
      else if (es == 8) generic_kernel<8, STREAM == 3><<<grid, block, 0, stream>>>(b);
      else generic_kernel<16, STREAM == 3><<<grid, block, 0, stream>>>(b);
      break;
  }
  CD_CHECK_HIP(hipGetLastError());
}

char g_last_kernel[96] = "";

void launchBatch(MoveClass cls, int variant, int stream_access, bool swizzle, bool window, int es, const Batch& b,
                 unsigned int blocks, hipStream_t stream) {
  // what ran last, in the words of the templates above (bench.py reports its dominant kernel from here)
  if (cls == MOVE_ROWS_VEC)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%d,%d>", window ? "rows_shifted_kernel" : "rows_kernel", variant,
             stream_access == 3 ? 3 : (stream_access >= 1 ? 1 : 0));
  else if (cls == MOVE_TRANSPOSE && window)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "transpose_window_kernel<%d,%d,%d,%d,%d>", es, variant % 100,
             es == 16 ? 32 : (variant >= 100 ? 128 : 64), es == 16 ? 32 : (es == 4 ? 128 : 64), stream_access);
  else if (cls == MOVE_TRANSPOSE)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "transpose_kernel<%d,%d,%d,%d,%d,%s>", es, variant % 100,
             es == 16 ? 32 : (variant == 204 ? 128 : 64), es == 16 ? 32 : (variant == 304 ? 128 : 64), stream_access,
             swizzle ? "true" : "false");
  else
    snprintf(g_last_kernel, sizeof(g_last_kernel), "generic_kernel<%d,%s>", es, stream_access == 3 ? "true" : "false");
  if (cls == MOVE_TRANSPOSE && window) {
    const bool wide = variant >= 100;
    if (stream_access == 3) launchWindowT<3>(variant % 100, es, b, blocks, stream, wide);
    else if (stream_access == 4 || stream_access == 2) launchWindowT<4>(variant % 100, es, b, blocks, stream, wide);
    else launchWindowT<0>(variant % 100, es, b, blocks, stream, wide);
    return;
  }
  if (stream_access == 4) {  // cached loads + streaming stores (misaligned sources)
    if (swizzle) launchBatchT<4, true>(cls, variant, es, b, blocks, stream, window);
    else launchBatchT<4, false>(cls, variant, es, b, blocks, stream, window);
    return;
  }
  // (mode 1 -- non-temporal loads, cached stores -- is a tuning mode of the transposes: its instantiations exist in tuning
  // builds only; row copies reach their streaming kernels through mode 2)
  if (swizzle) {
    if (stream_access == 3) launchBatchT<3, true>(cls, variant, es, b, blocks, stream, window);
    else if (stream_access == 2) launchBatchT<2, true>(cls, variant, es, b, blocks, stream, window);
#ifdef CUDECOMP_TUNING_VARIANTS
    else if (stream_access == 1) launchBatchT<1, true>(cls, variant, es, b, blocks, stream, window);
#endif
    else launchBatchT<0, true>(cls, variant, es, b, blocks, stream, window);
  } else {
    if (stream_access == 3) launchBatchT<3, false>(cls, variant, es, b, blocks, stream, window);
    else if (stream_access == 2) launchBatchT<2, false>(cls, variant, es, b, blocks, stream, window);
#ifdef CUDECOMP_TUNING_VARIANTS
    else if (stream_access == 1) launchBatchT<1, false>(cls, variant, es, b, blocks, stream, window);
#endif
    else launchBatchT<0, false>(cls, variant, es, b, blocks, stream, window);
  }
}

}  // namespace

const char* lastKernelName() { return g_last_kernel; }

void launchMoves(const Move3D* moves, int n, void* const bufs[3], int es, hipStream_t stream,
                 const KernelTuning* tuning, KernelStats* stats, void* const* dst_base_override) {
  const bool remote = dst_base_override != nullptr;
  if (es != 4 && es != 8 && es != 16) CD_INTERNAL_ERROR("unsupported element size");
  std::vector<Classified> cs;
  cs.reserve(n);
  for (int i = 0; i < n; ++i) {
    if (moves[i].elements() == 0) continue;
    cs.push_back(classify(moves[i], bufs, es, tuning, dst_base_override ? dst_base_override[i] : nullptr, remote));
  }
  // moves of one phase are independent, so they may be regrouped by kernel flavour
  std::vector<bool> done(cs.size(), false);
  for (size_t i = 0; i < cs.size(); ++i) {
    if (done[i]) continue;
    Batch b{};
    unsigned long long blocks = 0;
    for (size_t j = i; j < cs.size() && b.n < kMaxBatch; ++j) {
      if (done[j] || cs[j].cls != cs[i].cls || cs[j].variant != cs[i].variant || cs[j].stream != cs[i].stream ||
          cs[j].swizzle != cs[i].swizzle || cs[j].window != cs[i].window)
        continue;
      if (blocks + cs[j].blocks > 0x7fffffffULL) {
        if (b.n == 0) CD_NOT_SUPPORTED("single block move too large for one launch");
        break;
      }
      b.first_block[b.n] = (unsigned int)blocks;
      b.m[b.n] = cs[j].dm;
      b.p0[b.n] = cs[j].p0;
      b.p1[b.n] = cs[j].p1;
      b.t0[b.n] = cs[j].t0;
      b.t1[b.n] = cs[j].t1;
      blocks += cs[j].blocks;
      if (stats) stats->elements[cs[j].cls] += cs[j].elements;
      ++b.n;
      done[j] = true;
    }
    b.first_block[b.n] = (unsigned int)blocks;
    // Sibling row copies of one phase (the P chunks of an unpack, say) each touch one slice of every destination row:
    // run one after the other, a 2-KiB slice of every 8-KiB row keeps part of the memory channels idle.  Served round
    // robin, the workgroups in flight cover whole rows (C3 per-rank unpacks: 0.43-0.47 -> 0.35 ms, r02_tuning.md).
    // Transposes keep their XCD-contiguous tile walk (interleaving them measured slightly slower).
    const bool il_local = cs[i].cls != MOVE_TRANSPOSE && (!tuning || tuning->interleave_rows != 0);
    if ((dst_base_override || il_local) && b.n > 1) {
      unsigned long long widest = 0;
      for (int k = 0; k < b.n; ++k) widest = std::max<unsigned long long>(widest, b.first_block[k + 1] - b.first_block[k]);
      if (widest * b.n <= 0x7fffffffULL) {
        b.interleave = 1;
        blocks = widest * b.n;
      }
    }
    launchBatch(cs[i].cls, cs[i].variant, cs[i].stream, cs[i].swizzle, cs[i].window, es, b, (unsigned int)blocks, stream);
    if (stats) stats->launches[cs[i].cls] += 1;
  }
}

}  // namespace cudecomp
