// kernels_rows.hip -- row copies (fastest dim contiguous on both sides) and the generic element-wise fallback: hand-written
// gfx950 (CDNA4 / MI355X) data-movement kernels, one code object (see kernels_dev.h for why there are several).
//
// These replace the batched strided 3-D copy kernel of NVIDIA/cuDecomp (reference include/internal/cudecomp_kernels.cuh:125-180:
// one element per thread per iteration, two 64-bit div/mod pairs per element, no vector access):
//   rows_kernel<VB>       fastest dim contiguous on both sides.  Each lane moves VB = 16 (8, 4) bytes, a 256-thread workgroup
//                         keeps 4 vectors per lane (16 KiB) in flight, lanes run along the row so that every wavefront touches
//                         1 KiB contiguous segments.  No per-element index math: one (row, plane) decode per WORKGROUP.
//   rows_shifted_kernel   the same for destination rows off the 64-byte grid.
//   rows_dense_kernel     the same when, in addition, the cells between consecutive destination rows belong to the move
//                         (whole interior rows of a halo-carrying pencil): whole cache lines across the row ends.
//   generic_kernel<ES>    degenerate shapes (no unit stride on one side, 1-element rows).
// Pure data movement: no MFMA; the bound is HBM (8 TB/s spec, ~6.3 TB/s achievable copy rate).
#include "kernels_dev.h"

#include "errors.h"

namespace cudecomp {
namespace kern {
namespace {

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

// ---------------------------------------------------------------------------------------------
// rows_dense_kernel: row copy onto a halo-carrying pencil whose rows the move covers WHOLE (Move3D::dst_row_pitch).
// What costs on such destinations is not the misalignment but every 128-byte line that is only partly written -- the two
// lines at the ends of each row, about four line times each (profiles/r05_tuning.md section 3: 8 GiB onto rows of 8192 B at a
// pitch of 8208 B 3.25 ms against 2.85 ms aligned).  Here the lanes walk the destination's LINEAR address space on the 64-byte
// grid, across the row boundaries of a plane: every store of the body is a whole aligned 16-byte vector, every wavefront
// instruction writes 1 KiB of whole lines.  The few bytes between the end of one row and the start of the next (halo /
// padding cells of the same pencil, nobody else's during the operation) are read from the destination and written back
// unchanged.  Nothing is read or written below the first byte of a plane's first row or above the last byte of its last row:
// the two ends of a plane's span are copied in masked 4-byte pieces.  (Probe: scripts/tune/partial_probe.hip "dense": 2.83-2.87 ms.)
// e[0] = row length in BYTES, e[1] = rows per plane, e[2] = planes; ss/ds[1], [2] in bytes, ds[1] <= ds[2]; t0 = workgroups per
// plane; a workgroup covers kDenseBytes of the span.
// ---------------------------------------------------------------------------------------------
constexpr int kDenseBytes = kThreads * 16 * kRowsUnroll;

template <int STREAM>
__global__ __launch_bounds__(kThreads) void rows_dense_kernel(const Batch b) {
  using V = Bytes<16>;
  constexpr int POLICY = STREAM >= 1 ? ST_STREAM : ST_CACHED;
  int mi;
  unsigned int lb;
  if (!locate(b, blockIdx.x, mi, lb)) return;
  const DevMove& m = b.m[mi];
  const unsigned int per_plane = b.t0[mi];
  const long long plane = lb / per_plane;
  const long long chunk = lb % per_plane;
  const long long row_bytes = m.e[0], pitch = m.ds[1], spitch = m.ss[1];
  const char* __restrict__ s = m.src + plane * m.ss[2];
  char* __restrict__ d0 = m.dst + plane * m.ds[2];  // first byte of the plane's first row
  const long long shift = (long long)(reinterpret_cast<uintptr_t>(d0) & 63);
  const long long span = (m.e[1] - 1) * pitch + row_bytes;  // first byte of the first row .. last byte of the last row
  const double inv_pitch = 1.0 / (double)pitch;

  V v[kRowsUnroll];
  long long pos[kRowsUnroll];
  int kind[kRowsUnroll];  // 0 nothing, 1 whole vector, 2 an end of the span (masked pieces)
#pragma unroll
  for (int u = 0; u < kRowsUnroll; ++u) {
    const long long p = ((chunk * kRowsUnroll + u) * kThreads + threadIdx.x) * 16 - shift;  // byte offset from d0, 16-byte aligned address
    pos[u] = p;
    kind[u] = 0;
    if (p >= span || p + 16 <= 0) continue;
    // row and offset inside the row's pitch of byte max(p, 0): quotient by reciprocal, off by one at most
    const long long pc = p < 0 ? 0 : p;
    long long r = (long long)((double)pc * inv_pitch);
    long long o = pc - r * pitch;
    if (o < 0) {
      --r;
      o += pitch;
    } else if (o >= pitch) {
      ++r;
      o -= pitch;
    }
    if (p >= 0 && o + 16 <= row_bytes) {  // inside one row: the body
      kind[u] = 1;
      v[u] = loadVec<(STREAM >= 1), 16>(s + r * spitch + o);
      continue;
    }
    // a vector that holds a row end, a piece of the gap or an end of the span: dword by dword -- from the source inside a
    // row, from the destination itself inside a gap
    kind[u] = (p >= 0 && p + 16 <= span) ? 1 : 2;
    o -= pc - p;  // offset of byte p (negative only in front of the first row)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long pp = p + 4 * k;
      unsigned int w = 0;
      if (pp >= 0 && pp < span) {
        long long rr = r, oo = o + 4 * k;
        if (oo >= pitch) {
          ++rr;
          oo -= pitch;
        }
        w = oo < row_bytes ? *reinterpret_cast<const unsigned int*>(s + rr * spitch + oo) : *reinterpret_cast<const unsigned int*>(d0 + pp);
      }
      v[u][k] = w;
    }
  }
#pragma unroll
  for (int u = 0; u < kRowsUnroll; ++u) {
    if (kind[u] == 1) {
      storeVec<POLICY, 16>(d0 + pos[u], v[u]);
    } else if (kind[u] == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long pp = pos[u] + 4 * k;
        if (pp >= 0 && pp < span) storeVec<POLICY, 4>(d0 + pp, v[u][k]);
      }
    }
  }
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

}  // namespace
}  // namespace kern

using namespace kern;

int rowsDenseBytesPerBlock() { return kern::kDenseBytes; }

void launchRowsBatch(int mode, int vb, int stream_access, const Batch& b, unsigned int blocks, hipStream_t stream) {
  const dim3 grid(blocks), block(kThreads);
  const int rs = stream_access == 3 ? 3 : (stream_access >= 1 ? 1 : 0);  // (4 only occurs for transposes)
  const bool shifted = mode == 1;
  if (mode == 2) {  // local destinations only (kernels.cc classify())
    if (rs == 1) rows_dense_kernel<1><<<grid, block, 0, stream>>>(b);
    else rows_dense_kernel<0><<<grid, block, 0, stream>>>(b);
    CD_CHECK_HIP(hipGetLastError());
    return;
  }
#define CD_ROWS(K, VB)                                                    \
  do {                                                                    \
    if (rs == 3) K<VB, 3><<<grid, block, 0, stream>>>(b);                 \
    else if (rs == 1) K<VB, 1><<<grid, block, 0, stream>>>(b);            \
    else K<VB, 0><<<grid, block, 0, stream>>>(b);                         \
  } while (0)
  if (shifted) {
    if (vb == 16) CD_ROWS(rows_shifted_kernel, 16);
    else if (vb == 8) CD_ROWS(rows_shifted_kernel, 8);
    else CD_ROWS(rows_shifted_kernel, 4);
  } else {
    if (vb == 16) CD_ROWS(rows_kernel, 16);
    else if (vb == 8) CD_ROWS(rows_kernel, 8);
    else CD_ROWS(rows_kernel, 4);
  }
#undef CD_ROWS
  CD_CHECK_HIP(hipGetLastError());
}

void launchGenericBatch(int es, bool remote, const Batch& b, unsigned int blocks, hipStream_t stream) {
  const dim3 grid(blocks), block(kThreads);
  if (es == 4) {
    if (remote) generic_kernel<4, true><<<grid, block, 0, stream>>>(b);
    else generic_kernel<4, false><<<grid, block, 0, stream>>>(b);
  } else if (es == 8) {
    if (remote) generic_kernel<8, true><<<grid, block, 0, stream>>>(b);
    else generic_kernel<8, false><<<grid, block, 0, stream>>>(b);
  } else {
    if (remote) generic_kernel<16, true><<<grid, block, 0, stream>>>(b);
    else generic_kernel<16, false><<<grid, block, 0, stream>>>(b);
  }
  CD_CHECK_HIP(hipGetLastError());
}

}  // namespace cudecomp
