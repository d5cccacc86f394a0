// partial_probe.hip -- why does a copy onto rows that are off the 64-byte grid lose 11-15 % (round 5)?
//
// A dense 8 GiB row copy, rows of 8192 B at a pitch of 8208 B (a 1024-wide fp64 pencil with a halo of 1), destination rows
// starting 8 B past a 64-byte boundary.  Lanes are laid out on the destination's 64-byte grid (as rows_shifted_kernel does);
// the variants differ ONLY in what happens to the two partial units at the ends of every row (1.6 % of the units):
//   partial   the partial units are written as they are (8-byte pieces)                     = what the library does
//   skip      the partial units are not written at all (wrong result; shows their cost)
//   full      whole units are written at the row ends too (the bytes outside the row -- halo cells of the pencil -- are read
//             from the destination first and written back unchanged)
//   aligned   pitch 8192, no offset: the ceiling
// Also: source aligned vs. source shifted like the destination; "dense": the linear walk across row ends (whole lines, gap bytes
// rewritten unchanged) that became rows_dense_kernel; and the library's own shifted / dense kernels on the same buffers.
//   build:  make -C ../../cudecomp_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../cudecomp_amd/csrc -I../../include -c partial_probe.hip -o pp.o \
//             && hipcc --offload-arch=gfx950 pp.o ../../cudecomp_amd/build/{kernels,plan,decomp}.o ../../cudecomp_amd/build/kernels_*.hip.o -o partial_probe
#include <hip/hip_runtime.h>

#include "kernels.h"  // the library's dispatch (linked from cudecomp_amd/build/*.o): rows_shifted_kernel / rows_dense_kernel themselves

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e = (x);                                                \
    if (e != hipSuccess) {                                             \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(4))) u32x4_g;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((aligned(4))) u32x2_g;

// MODE 0 partial, 1 skip, 2 full.  One workgroup = 4 rows x 4 passes; a row has `units` 64-byte units on the destination grid
// (the first and last may be partial); lane = one 16-byte quarter of a unit.
template <int MODE>
__global__ __launch_bounds__(256) void copy_k(const char* __restrict__ src, char* __restrict__ dst, long long rows, long long pitch,
                                              long long row_bytes, long long doff, long long soff) {
  const long long r0 = (long long)blockIdx.x * 2;
  for (int rr = 0; rr < 2; ++rr) {
    const long long r = r0 + rr;
    if (r >= rows) return;
    char* drow = dst + r * pitch + doff;           // first byte of the row
    const char* srow = src + r * pitch + soff;
    const long long shift = (long long)(reinterpret_cast<uintptr_t>(drow) & 63);
    const long long nvec = (shift + row_bytes + 15) / 16;  // 16-byte pieces from the unit boundary below the row start
    u32x4 v[3];
    long long off[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const long long q = threadIdx.x + p * 256;
      off[p] = q * 16 - shift;  // byte offset inside the row
      if (q < nvec && off[p] >= 0 && off[p] + 16 <= row_bytes) v[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_g*>(srow + off[p]));
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const long long q = threadIdx.x + p * 256;
      if (q >= nvec) continue;
      if (off[p] >= 0 && off[p] + 16 <= row_bytes) {
        __builtin_nontemporal_store(v[p], reinterpret_cast<u32x4_g*>(drow + off[p]));
      } else if (MODE == 0) {  // 8-byte pieces inside the row
        for (int k = 0; k < 2; ++k) {
          const long long o = off[p] + 8 * k;
          if (o >= 0 && o + 8 <= row_bytes) *reinterpret_cast<u32x2_g*>(drow + o) = *reinterpret_cast<const u32x2_g*>(srow + o);
        }
      }
    }
    if (MODE == 2) {
      // the two end UNITS as whole 64-byte writes by lanes 0..3 / 4..7: inside the row from the source, outside from the
      // destination's present content
      const int l = threadIdx.x;
      if (l < 8) {
        const long long ubase = (l < 4) ? -shift : ((shift + row_bytes) / 64) * 64 - shift;  // unit start relative to row start
        const long long o = ubase + (l & 3) * 16;
        u32x4 w;
        unsigned int* wp = reinterpret_cast<unsigned int*>(&w);
        for (int k = 0; k < 4; ++k) {
          const long long b = o + 4 * k;
          wp[k] = (b >= 0 && b + 4 <= row_bytes) ? *reinterpret_cast<const unsigned int*>(srow + b) : *reinterpret_cast<const unsigned int*>(drow + b);
        }
        const bool partial_unit = (l < 4) ? shift != 0 : ((shift + row_bytes) % 64) != 0;
        if (partial_unit) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(drow + o));
      }
    }
  }
}


// MODE 3 "dense" (round 5, last GPU calls): the lanes walk the destination's LINEAR address space on the 64-byte grid, across
// row boundaries: every store is a whole aligned vector and every wavefront instruction writes 1 KiB of whole lines.  Bytes of
// the gaps between rows (halo cells of the same pencil) are read from the destination and written back unchanged; nothing is
// touched below the first row's first byte or above the last row's last byte (masked pieces there).  spitch: the source's own
// row pitch (8192 = a dense receive area, the unpack case).
template <int BAR>
__global__ __launch_bounds__(256) void dense_k(const char* __restrict__ src, char* __restrict__ dst, long long rows, long long pitch,
                                               long long spitch, long long row_bytes, long long doff, long long soff, double inv_pitch) {
  char* d0 = dst + doff;  // first byte of row 0
  const long long shift = (long long)(reinterpret_cast<uintptr_t>(d0) & 63);
  const long long span = (rows - 1) * pitch + row_bytes;  // bytes from row 0's first to the last row's last byte
  u32x4 v[4];
  long long pos[4];
  int kind[4];  // 0 nothing, 1 whole vector, 2 mixed (v holds the merged vector), 3 edge of the span (pieces)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    // mapping (BAR >> 1): 0 a workgroup covers 16 KiB contiguous (lane stride 4 KiB); 1 two workgroups interleave 4-KiB pieces of a
    // 32-KiB chunk (lane stride 8 KiB, as rows_kernel on 8-KiB rows); 2 four workgroups interleave (lane stride 16 KiB)
    const long long bx = blockIdx.x;
    const long long q = (BAR >> 1) == 0 ? (bx * 4 + u) * 256 + threadIdx.x
                        : ((BAR >> 1) == 1 ? ((bx / 2) * 8 + u * 2 + (bx % 2)) * 256 + threadIdx.x : ((bx / 4) * 16 + u * 4 + (bx % 4)) * 256 + threadIdx.x);
    const long long p = q * 16 - shift;
    pos[u] = p;
    kind[u] = 0;
    if (p >= span || p + 16 <= 0) continue;
    long long r = (long long)((double)(p < 0 ? 0 : p) * inv_pitch);
    long long o = p - r * pitch;
    if (o < 0) { --r; o += pitch; } else if (o >= pitch) { ++r; o -= pitch; }
    if (p >= 0 && o + 16 <= row_bytes) {
      kind[u] = 1;
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_g*>(src + r * spitch + soff + o));
    } else {
      kind[u] = (p >= 0 && p + 16 <= span) ? 2 : 3;
      unsigned int* wp = reinterpret_cast<unsigned int*>(&v[u]);
      for (int k = 0; k < 4; ++k) {
        const long long pp = p + 4 * k;
        if (pp < 0 || pp >= span) { wp[k] = 0; continue; }
        long long rr = r, oo = o + 4 * k;
        if (oo >= pitch) { ++rr; oo -= pitch; }
        wp[k] = oo < row_bytes ? *reinterpret_cast<const unsigned int*>(src + rr * spitch + soff + oo) : *reinterpret_cast<const unsigned int*>(d0 + pp);
      }
    }
  }
  if (BAR & 1) __syncthreads();  // (mode 5: all four wavefronts have their loads back before anybody stores)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (kind[u] == 1 || kind[u] == 2) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(d0 + pos[u]));
    else if (kind[u] == 3) {
      const unsigned int* wp = reinterpret_cast<const unsigned int*>(&v[u]);
      for (int k = 0; k < 4; ++k) {
        const long long pp = pos[u] + 4 * k;
        if (pp >= 0 && pp < span) *reinterpret_cast<unsigned int*>(d0 + pp) = wp[k];
      }
    }
  }
}

// MODE 4 "dense2": the dense walk with ALIGNED loads -- each lane fetches the aligned 16-byte vector below its (misaligned)
// source bytes, takes the next one from its neighbour lane (ds_bpermute; the wavefront's last lane from the next wavefront of
// the workgroup through LDS) and funnel-shifts the two by the misalignment; lanes at row ends keep the unaligned load.
__device__ __forceinline__ u32x4 shiftPair(const u32x4& L, const u32x4& H, int d) {  // dwords d .. d+3 of (L, H)
  u32x4 r;
  r.x = d == 0 ? L.x : (d == 1 ? L.y : (d == 2 ? L.z : L.w));
  r.y = d == 0 ? L.y : (d == 1 ? L.z : (d == 2 ? L.w : H.x));
  r.z = d == 0 ? L.z : (d == 1 ? L.w : (d == 2 ? H.x : H.y));
  r.w = d == 0 ? L.w : (d == 1 ? H.x : (d == 2 ? H.y : H.z));
  return r;
}
__global__ __launch_bounds__(256) void dense2_k(const char* __restrict__ src, char* __restrict__ dst, long long rows, long long pitch,
                                                long long spitch, long long row_bytes, long long doff, long long soff, double inv_pitch) {
  __shared__ u32x4 xch[4][4];
  char* d0 = dst + doff;
  const long long shift = (long long)(reinterpret_cast<uintptr_t>(d0) & 63);
  const long long span = (rows - 1) * pitch + row_bytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 v[4], L[4];
  long long pos[4];
  const char* addr[4];
  int kind[4];  // 0 nothing, 1 whole vector (v final), 2 span edge, 4 body with aligned load (needs the shift), 5 same, successor not usable
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long q = ((long long)blockIdx.x * 4 + u) * 256 + threadIdx.x;
    const long long p = q * 16 - shift;
    pos[u] = p;
    kind[u] = 0;
    L[u] = u32x4{0, 0, 0, 0};
    addr[u] = src;
    if (p >= span || p + 16 <= 0) continue;
    long long r = (long long)((double)(p < 0 ? 0 : p) * inv_pitch);
    long long o = p - r * pitch;
    if (o < 0) { --r; o += pitch; } else if (o >= pitch) { ++r; o -= pitch; }
    if (p >= 0 && o + 16 <= row_bytes) {
      const char* a = src + r * spitch + soff + o;
      const int sh = (int)(reinterpret_cast<uintptr_t>(a) & 15);
      addr[u] = a;
      if (sh == 0) {
        kind[u] = 1;
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a));
        L[u] = v[u];
      } else {
        kind[u] = (o + 32 <= row_bytes) ? 4 : 5;
        L[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a - sh));
      }
    } else {
      kind[u] = (p >= 0 && p + 16 <= span) ? 1 : 2;
      unsigned int* wp = reinterpret_cast<unsigned int*>(&v[u]);
      for (int k = 0; k < 4; ++k) {
        const long long pp = p + 4 * k;
        if (pp < 0 || pp >= span) { wp[k] = 0; continue; }
        long long rr = r, oo = o + 4 * k;
        if (oo >= pitch) { ++rr; oo -= pitch; }
        wp[k] = oo < row_bytes ? *reinterpret_cast<const unsigned int*>(src + rr * spitch + soff + oo) : *reinterpret_cast<const unsigned int*>(d0 + pp);
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) xch[u][wave] = L[u];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    u32x4 H;
    H.x = __shfl_down(L[u].x, 1);
    H.y = __shfl_down(L[u].y, 1);
    H.z = __shfl_down(L[u].z, 1);
    H.w = __shfl_down(L[u].w, 1);
    if (kind[u] < 4) continue;
    const int sh = (int)(reinterpret_cast<uintptr_t>(addr[u]) & 15);
    bool have = kind[u] == 4;
    if (lane == 63 && have) {
      if (wave < 3) H = xch[u][wave + 1];
      else if (u < 3) H = xch[u + 1][0];
      else have = false;
    }
    if (!have) H = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(addr[u] - sh + 16));
    v[u] = shiftPair(L[u], H, sh >> 2);
    kind[u] = 1;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (kind[u] == 1) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(d0 + pos[u]));
    else if (kind[u] == 2) {
      const unsigned int* wp = reinterpret_cast<const unsigned int*>(&v[u]);
      for (int k = 0; k < 4; ++k) {
        const long long pp = pos[u] + 4 * k;
        if (pp >= 0 && pp < span) *reinterpret_cast<unsigned int*>(d0 + pp) = wp[k];
      }
    }
  }
}

template <int V2>
float timeDense(const char* src, char* dst, long long rows, long long pitch, long long spitch, long long row_bytes, long long doff, long long soff) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long long span = (rows - 1) * pitch + row_bytes + 64;
  const unsigned blocks = (unsigned)((span + 16383) / 16384);
  auto go = [&]() {
    if (V2 == 1) dense2_k<<<blocks, 256>>>(src, dst, rows, pitch, spitch, row_bytes, doff, soff, 1.0 / (double)pitch);
    else if (V2 == 2) dense_k<1><<<blocks, 256>>>(src, dst, rows, pitch, spitch, row_bytes, doff, soff, 1.0 / (double)pitch);
    else if (V2 == 3) dense_k<2><<<(blocks + 3) / 4 * 4, 256>>>(src, dst, rows, pitch, spitch, row_bytes, doff, soff, 1.0 / (double)pitch);
    else if (V2 == 4) dense_k<4><<<(blocks + 3) / 4 * 4, 256>>>(src, dst, rows, pitch, spitch, row_bytes, doff, soff, 1.0 / (double)pitch);
    else dense_k<0><<<blocks, 256>>>(src, dst, rows, pitch, spitch, row_bytes, doff, soff, 1.0 / (double)pitch);
  };
  go();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) go();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 10;
}
// the dense kernel's result, every byte: rows = source rows, gaps and everything outside the span = the destination's previous content
__global__ void fill_k(unsigned int* p, unsigned long long n, unsigned int salt) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
    p[i] = (unsigned int)(i * 2654435761u) ^ salt;
}
__global__ void check_k(const unsigned int* src, const unsigned int* dst, unsigned long long ndw, long long rows, long long pitch, long long spitch,
                        long long row_bytes, long long doff, long long soff, unsigned int salt, unsigned long long* bad) {
  unsigned long long mine = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < ndw; i += (unsigned long long)gridDim.x * blockDim.x) {
    const long long p = (long long)i * 4 - doff;
    unsigned int want = (unsigned int)(i * 2654435761u) ^ salt;  // untouched
    if (p >= 0) {
      const long long r = p / pitch, o = p % pitch;
      if (r < rows && o < row_bytes) want = src[(r * spitch + soff + o) / 4];
    }
    if (dst[i] != want) ++mine;
  }
  if (mine) atomicAdd(bad, mine);
}

template <int MODE>
float timeIt(const char* src, char* dst, long long rows, long long pitch, long long row_bytes, long long doff, long long soff) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const unsigned blocks = (unsigned)((rows + 1) / 2);
  copy_k<MODE><<<blocks, 256>>>(src, dst, rows, pitch, row_bytes, doff, soff);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) copy_k<MODE><<<blocks, 256>>>(src, dst, rows, pitch, row_bytes, doff, soff);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 10;
}

int main() {
  const long long rows = 1 << 20, row_bytes = 8192;
  const size_t cap = (size_t)rows * 8320 + 4096;
  char *src, *dst;
  CK(hipMalloc(&src, cap));
  CK(hipMalloc(&dst, cap));
  CK(hipMemset(src, 1, cap));
  CK(hipMemset(dst, 2, cap));
  const double bytes = 2.0 * rows * row_bytes;
  struct Case {
    const char* name;
    long long pitch, doff, soff;
  } cases[] = {{"aligned: pitch 8192, no offsets (ceiling)", 8192, 0, 0},
               {"pitch 8208, dst +8 B, src +8 B (halo pencil to halo pencil)", 8208, 8, 8},
               {"pitch 8208, dst +8 B, src aligned rows (pitch 8192)", 8208, 8, -1},
               {"pitch 8256 (64-B multiple), dst +8 B, src +8 B", 8256, 8, 8},
               {"pitch 8208, dst +0 (rows drift over the grid, first row aligned)", 8208, 0, 0},
               {"pitch 8320 = 65 lines: rows line-aligned, 128-B holes between rows", 8320, 0, 0},
               {"pitch 8256: rows unit-aligned (every other row starts mid-line), 64-B holes", 8256, 0, 0},
               {"pitch 8192, dst +8 B, src +8 B: dense, everything 8 B off the grid (no holes)", 8192, 8, 8},
               {"pitch 8192, dst +64 B, src +64 B: dense, half a line off", 8192, 64, 64},
               {"pitch 8192, dst +0, src +8 B: only the loads off the grid", 8192, 0, 8},
               {"pitch 8192, dst +8, src +0 B: only the stores off the grid (lanes on the dst grid)", 8192, 8, 0}};
  for (auto& c : cases) {
    printf("== %s\n", c.name);
    // (soff -1: the source uses its own dense pitch; emulated by giving the source the same pitch but offset 0 -- the loads
    // are then aligned only for every fourth row; good enough to separate the load side from the store side)
    const long long soff = c.soff < 0 ? 0 : c.soff;
    float a = timeIt<0>(src, dst, rows, c.pitch, row_bytes, c.doff, soff);
    float b = timeIt<1>(src, dst, rows, c.pitch, row_bytes, c.doff, soff);
    float f = timeIt<2>(src, dst, rows, c.pitch, row_bytes, c.doff, soff);
    printf("  partial units written as pieces : %.3f ms %6.0f GB/s\n", a, bytes / a / 1e6);
    printf("  partial units skipped           : %.3f ms %6.0f GB/s\n", b, bytes / b / 1e6);
    printf("  end units written whole (RMW)   : %.3f ms %6.0f GB/s\n", f, bytes / f / 1e6);
    {  // dense: checked byte by byte first (pattern data), then timed; source with the same pitch, and with its own dense pitch
      for (int v2 = 0; v2 < 5; ++v2)
      for (long long spitch : {c.pitch, (long long)8192}) {
        const unsigned long long ndw = cap / 4;
        fill_k<<<8192, 256>>>(reinterpret_cast<unsigned int*>(src), ndw, 0x12345678u);
        fill_k<<<8192, 256>>>(reinterpret_cast<unsigned int*>(dst), ndw, 0x9abcdef0u);
        const unsigned nblk = (unsigned)(((rows - 1) * c.pitch + row_bytes + 64 + 16383) / 16384);
        if (v2 == 1) dense2_k<<<nblk, 256>>>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff, 1.0 / (double)c.pitch);
        else if (v2 == 2) dense_k<1><<<nblk, 256>>>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff, 1.0 / (double)c.pitch);
        else if (v2 == 3) dense_k<2><<<(nblk + 3) / 4 * 4, 256>>>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff, 1.0 / (double)c.pitch);
        else if (v2 == 4) dense_k<4><<<(nblk + 3) / 4 * 4, 256>>>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff, 1.0 / (double)c.pitch);
        else dense_k<0><<<nblk, 256>>>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff, 1.0 / (double)c.pitch);
        unsigned long long* bad;
        CK(hipMalloc(&bad, 8));
        CK(hipMemset(bad, 0, 8));
        check_k<<<8192, 256>>>(reinterpret_cast<const unsigned int*>(src), reinterpret_cast<const unsigned int*>(dst), ndw, rows, c.pitch, spitch,
                               row_bytes, c.doff, soff, 0x9abcdef0u, bad);
        unsigned long long nbad = 1;
        CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
        CK(hipFree(bad));
        float dn = v2 == 1 ? timeDense<1>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff)
                           : (v2 == 2 ? timeDense<2>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff)
                           : (v2 == 3 ? timeDense<3>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff)
                           : (v2 == 4 ? timeDense<4>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff) : timeDense<0>(src, dst, rows, c.pitch, spitch, row_bytes, c.doff, soff))));
        printf("  dense linear walk%s, gaps RMW, source pitch %lld : %.3f ms %6.0f GB/s   (%llu wrong dwords)\n", v2 == 1 ? " + ALIGNED loads (lane shift)" : (v2 == 2 ? " + BARRIER between loads and stores" : (v2 == 3 ? " two workgroups interleaved (lane stride 8 KiB)" : (v2 == 4 ? " four workgroups interleaved (lane stride 16 KiB)" : ""))),
               spitch, dn, bytes / dn / 1e6, nbad);
      }
      CK(hipMemset(src, 1, cap));
      CK(hipMemset(dst, 2, cap));
    }
  }
  {  // the LIBRARY's kernels on the same buffers (pattern data): plain / shifted / dense, source pitch 8192 and 8208
    using namespace cudecomp;
    fill_k<<<8192, 256>>>(reinterpret_cast<unsigned int*>(src), cap / 4, 0x12345678u);
    fill_k<<<8192, 256>>>(reinterpret_cast<unsigned int*>(dst), cap / 4, 0x9abcdef0u);
    for (long long sp : {1024ll, 1026ll})
      for (long long dp : {1024ll, 1026ll, 1028ll})
        for (int whole = 0; whole < 2; ++whole) {
          Move3D m;
          m.src_buf = BUF_IN;
          m.dst_buf = BUF_OUT;
          m.extent[0] = 1024; m.extent[1] = rows; m.extent[2] = 1;
          m.ss[0] = 1; m.ss[1] = sp; m.ss[2] = 0;
          m.ds[0] = 1; m.ds[1] = dp; m.ds[2] = 0;
          m.dst_off = dp == 1024 ? 0 : 1;
          m.dst_row_pitch = whole ? dp : 0;
          void* bufs[3] = {src, dst, nullptr};
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0));
          CK(hipEventCreate(&e1));
          KernelTuning kt;
          launchMoves(&m, 1, bufs, 8, nullptr, &kt);
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          for (int i = 0; i < 10; ++i) launchMoves(&m, 1, bufs, 8, nullptr, &kt);
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          ms /= 10;
          printf("library: source pitch %lld B, destination pitch %lld B (+%lld B), whole rows %d: %-28s %.3f ms %6.0f GB/s\n", sp * 8, dp * 8,
                 (long long)m.dst_off * 8, whole, lastKernelName(), ms, bytes / ms / 1e6);
        }
  }
  return 0;
}
