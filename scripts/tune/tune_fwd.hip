// tune_fwd.hip -- round 5: the forward hops of the 1024^3 axis-contiguous cycle (destination rows 8 MiB apart) against the
// inverse ones (source rows 8 MiB apart), with the LIBRARY's own tile code (this file includes csrc/kernels_tile.h, so every
// variant here is the shipped transposeTile / transposeTilePadded with other template arguments or another tile walk; nothing
// is added to the library's device code).  Not part of the product.
//
//   build:  make -C ../../cudecomp_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../cudecomp_amd/csrc -I../../include tune_fwd.hip \
//               ../../cudecomp_amd/build/{kernels,plan,decomp}.o ../../cudecomp_amd/build/kernels_*.hip.o -o tune_fwd
//   run:    ./tune_fwd [8|16] [reps]
//
// Sections: (1) tile shapes x access modes x tile walks, per direction; (2) the same launch isolated (host sync between
// launches) and sustained (back to back, events between launches): where the time between kernels goes.
#include "kernels_tile.h"
#include "kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace cudecomp;
using namespace cudecomp::kern;

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e = (x);                                                \
    if (e != hipSuccess) {                                             \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

// A tile walk as a mixed-radix number: workgroup index (after the XCD remap) -> digits, least significant first; digit d has
// `size[d]` values and contributes value * mult_{i,j,k}[d] to the tile coordinates.
struct Walk {
  int n;
  int xcd;  // 1: XCD-contiguous runs (workgroup b -> XCD b % 8 gets a contiguous run of the sequence)
  unsigned size[6], mi[6], mj[6], mk[6];
};
struct TArgs {
  const char* src;
  char* dst;
  long long ei, ej, ek, sj, sk, di, dk;
  Walk w;
};

template <int ES, int VW, int TI, int TJ, int STREAM, bool SWZ>
__global__ __launch_bounds__(kThreads) void walk_kernel(const TArgs a) {
  using E = Bytes<ES>;
  __shared__ __attribute__((aligned(16))) E tile[SWZ ? TJ * TI : TJ * (TI + 1)];
  unsigned lt = blockIdx.x;
  if (a.w.xcd) {
    const unsigned per = gridDim.x >> 3;
    if (lt < (per << 3)) lt = (lt & 7u) * per + (lt >> 3);
  }
  unsigned bi = 0, bj = 0, k = 0;
  for (int d = 0; d < a.w.n; ++d) {
    const unsigned v = lt % a.w.size[d];
    lt /= a.w.size[d];
    bi += v * a.w.mi[d];
    bj += v * a.w.mj[d];
    k += v * a.w.mk[d];
  }
  const long long i0 = (long long)bi * TI, j0 = (long long)bj * TJ;
  const E* __restrict__ src = reinterpret_cast<const E*>(a.src) + (long long)k * a.sk;
  E* __restrict__ dst = reinterpret_cast<E*>(a.dst) + (long long)k * a.dk;
  if constexpr (SWZ) transposeTile<ES, VW, TI, TJ, STREAM, false>(tile, src, dst, i0, j0, a.ei, a.ej, a.sj, a.di, threadIdx.x);
  else transposeTilePadded<ES, VW, TI, TJ, STREAM, false>(tile, src, dst, i0, j0, a.ei, a.ej, a.sj, a.di, threadIdx.x);
}


// ---- phase 5 (round 5, last GPU call): the swizzled tile with (a) the store's cache bits spelled out, (b) persistent workgroups
// that fetch tile t+1 while tile t is written, (c) occupancy limits through dynamic LDS.  A copy of transposeTile's two phases
// with the store and the tile loop made pluggable -- harness only.
// SP: bit 0 = sc0, bit 1 = sc1, bit 2 = nt; 8 = the compiler's own non-temporal store (what the library ships)
template <int SP> __device__ __forceinline__ void store16(void* p, const u32x4& v) {
  if constexpr (SP == 8) __builtin_nontemporal_store(v, static_cast<u32x4_g*>(p));
  else if constexpr (SP == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SP == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SP == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SP == 4) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SP == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SP == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int TI, int TJ> struct TileMap {  // 8-byte elements, 16-byte vectors (VW = 2), as transposeTile
  static constexpr int VW = 2, G = TI / VW, TPR = TI / VW, RPP = kThreads / TPR, NP = TJ / RPP;
  static constexpr int TPO = TJ / VW, BPO = kThreads / TPO, NPO = TI / (BPO * VW);
};

template <int TI, int TJ, bool NTLOAD>
__device__ __forceinline__ void tileLoad(u32x4* regs, const u32x2* __restrict__ src, long long i0, long long j0,
                                         long long sj, int tid) {
  using M = TileMap<TI, TJ>;
  const int lg = tid % M::TPR, lj = tid / M::TPR;
  const u32x2* base = src + (j0 + lj) * sj + i0 + lg * M::VW;
#pragma unroll
  for (int p = 0; p < M::NP; ++p) regs[p] = loadVec<NTLOAD, 16>(base + (long long)(p * M::RPP) * sj);
}
template <int TI, int TJ>
__device__ __forceinline__ void tileToLds(u32x4* vtile, const u32x4* regs, int tid) {
  using M = TileMap<TI, TJ>;
  const int lg = tid % M::TPR, lj = tid / M::TPR;
#pragma unroll
  for (int p = 0; p < M::NP; ++p) {
    const int r = lj + p * M::RPP;
    vtile[r * M::G + (lg ^ ((r / M::VW) % M::G))] = regs[p];
  }
}
template <int TI, int TJ, int SP>
__device__ __forceinline__ void tileStore(const u32x4* vtile, u32x2* __restrict__ dst, long long i0, long long j0, long long di, int tid) {
  using M = TileMap<TI, TJ>;
  const int ljg = tid % M::TPO, lj = ljg * M::VW, lb = tid / M::TPO;
#pragma unroll
  for (int p = 0; p < M::NPO; ++p) {
    const int ig = lb + p * M::BPO;
    u32x4 in[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) in[v] = vtile[(lj + v) * M::G + (ig ^ (ljg % M::G))];
    u32x4 o0, o1;
    o0.xy = in[0].xy; o0.zw = in[1].xy;
    o1.xy = in[0].zw; o1.zw = in[1].zw;
    store16<SP>(dst + (i0 + ig * 2) * di + j0 + lj, o0);
    store16<SP>(dst + (i0 + ig * 2 + 1) * di + j0 + lj, o1);
  }
}

__device__ __forceinline__ void decodeWalk(const Walk& w, unsigned lt, unsigned& bi, unsigned& bj, unsigned& k) {
  bi = bj = k = 0;
  for (int d = 0; d < w.n; ++d) {
    const unsigned v = lt % w.size[d];
    lt /= w.size[d];
    bi += v * w.mi[d];
    bj += v * w.mj[d];
    k += v * w.mk[d];
  }
}

// NT consecutive tiles of the walk per workgroup (NT = 1: one tile, as the library); NT > 1 prefetches tile t+1 into registers
// before tile t is written out.
template <int TI, int TJ, int SP, int NT, bool NTLOAD>
__global__ __launch_bounds__(kThreads) void walk5_kernel(const TArgs a) {
  using M = TileMap<TI, TJ>;
  __shared__ __attribute__((aligned(16))) u32x2 tile[TJ * TI];
  u32x4* vtile = reinterpret_cast<u32x4*>(tile);
  unsigned wg = blockIdx.x;
  if (a.w.xcd) {
    const unsigned per = gridDim.x >> 3;
    if (wg < (per << 3)) wg = (wg & 7u) * per + (wg >> 3);
  }
  const int tid = threadIdx.x;
  unsigned bi, bj, k;
  decodeWalk(a.w, wg * NT, bi, bj, k);
  u32x4 regs[M::NP];
  tileLoad<TI, TJ, NTLOAD>(regs, reinterpret_cast<const u32x2*>(a.src) + (long long)k * a.sk, (long long)bi * TI, (long long)bj * TJ, a.sj, tid);
  // the first tile is peeled so that inside the loop the prefetched loads are always FOLLOWED by a tile's stores: the compiler's
  // s_waitcnt for the loads then leaves those stores in flight (merged with a store-free loop entry it waits for vmcnt(0))
  auto step = [&](int t) {
    tileToLds<TI, TJ>(vtile, regs, tid);
    __syncthreads();
    const unsigned ci = bi, cj = bj, ck = k;
    if (t + 1 < NT) {
      decodeWalk(a.w, wg * NT + t + 1, bi, bj, k);
      tileLoad<TI, TJ, NTLOAD>(regs, reinterpret_cast<const u32x2*>(a.src) + (long long)k * a.sk, (long long)bi * TI, (long long)bj * TJ, a.sj, tid);
    }
    tileStore<TI, TJ, SP>(vtile, reinterpret_cast<u32x2*>(a.dst) + (long long)ck * a.dk, (long long)ci * TI, (long long)cj * TJ, a.di, tid);
    if (t + 1 < NT) __syncthreads();
  };
  step(0);
#pragma unroll 1
  for (int t = 1; t < NT; ++t) step(t);
}

// phase 5 data: a position-dependent pattern in the source and a full comparison of the destination (every cell) after each
// variant -- the phase-5 kernels are new code, not the library's.
__global__ void fill_pattern_k(unsigned long long* p, unsigned long long n) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
    p[i] = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
}
__global__ void check_move_k(const unsigned long long* src, const unsigned long long* dst, long long ei, long long ej, long long ek, long long sj,
                             long long sk, long long di, long long dk, unsigned long long* bad) {
  const unsigned long long n = (unsigned long long)ei * ej * ek;
  unsigned long long mine = 0;
  for (unsigned long long t = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * blockDim.x) {
    const long long i = t % ei, j = (t / ei) % ej, k = t / (ei * ej);
    if (dst[i * di + j + k * dk] != src[i + j * sj + k * sk]) ++mine;
  }
  if (mine) atomicAdd(bad, mine);
}
static unsigned long long* g_bad = nullptr;

struct Shape {
  const char* name;
  long long ei, ej, ek, sj, sk, di, dk;
};

// walks: named orders of (i tiles, j tiles, k), optionally with k split into k_lo (KL values) and k_hi
static Walk makeWalk(const char* order, unsigned ti_n, unsigned tj_n, unsigned ek, unsigned KL, int xcd) {
  Walk w{};
  w.xcd = xcd;
  for (const char* c = order; *c; ++c) {
    const int d = w.n++;
    w.mi[d] = w.mj[d] = w.mk[d] = 0;
    if (*c == 'i') { w.size[d] = ti_n; w.mi[d] = 1; }
    else if (*c == 'j') { w.size[d] = tj_n; w.mj[d] = 1; }
    else if (*c == 'k') { w.size[d] = ek; w.mk[d] = 1; }          // whole k
    else if (*c == 'l') { w.size[d] = KL; w.mk[d] = 1; }          // k_lo
    else if (*c == 'h') { w.size[d] = ek / KL; w.mk[d] = KL; }    // k_hi
    else if (*c == 'a') { w.size[d] = KL; w.mi[d] = 1; }          // i_lo (KL tiles)
    else if (*c == 'b') { w.size[d] = ti_n / KL; w.mi[d] = KL; }  // i_hi
  }
  return w;
}

template <int ES, int VW, int TI, int TJ, int STREAM, bool SWZ>
float timeVariant(const Shape& s, const char* src, char* dst, const char* order, unsigned KL, int xcd, int reps) {
  TArgs a{src, dst, s.ei, s.ej, s.ek, s.sj, s.sk, s.di, s.dk, makeWalk(order, (unsigned)(s.ei / TI), (unsigned)(s.ej / TJ), (unsigned)s.ek, KL, xcd)};
  const unsigned blocks = (unsigned)(s.ei / TI) * (unsigned)(s.ej / TJ) * (unsigned)s.ek;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  walk_kernel<ES, VW, TI, TJ, STREAM, SWZ><<<blocks, kThreads>>>(a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) walk_kernel<ES, VW, TI, TJ, STREAM, SWZ><<<blocks, kThreads>>>(a);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / reps;
}

static double g_bytes = 0;
static size_t g_buf_bytes = 0;
#define ROW(ES, VW, TI, TJ, STREAM, SWZ, ORDER, KL, XCD)                                                                  \
  {                                                                                                                        \
    float f = timeVariant<ES, VW, TI, TJ, STREAM, SWZ>(shapes[0], src, dst, ORDER, KL, XCD, reps);                          \
    float b = timeVariant<ES, VW, TI, TJ, STREAM, SWZ>(shapes[1], src, dst, ORDER, KL, XCD, reps);                          \
    printf("| %3d x %3d | %d | %-5s | %-6s KL=%-2d xcd=%d | %.3f (%.3f) | %.3f (%.3f) |\n", TI, TJ, STREAM, SWZ ? "swz" : "pad", \
           ORDER, KL, XCD, f, g_bytes / f / 8e9, b, g_bytes / b / 8e9);                                                     \
    fflush(stdout);                                                                                                        \
  }


static bool checkMove(const Shape& s, const char* src, const char* dst) {
  if (!g_bad) CK(hipMalloc(&g_bad, 8));
  CK(hipMemset(g_bad, 0, 8));
  check_move_k<<<8192, 256>>>(reinterpret_cast<const unsigned long long*>(src), reinterpret_cast<const unsigned long long*>(dst), s.ei, s.ej, s.ek,
                              s.sj, s.sk, s.di, s.dk, g_bad);
  unsigned long long bad = 1;
  CK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
  return bad == 0;
}

static long long g_far_pad = 0;  // elements added to the far stride (fwd: destination rows, bwd: source rows): DRAM channel aliasing probe
static unsigned g_dyn_lds = 0;  // extra dynamic LDS per workgroup (unused by the kernel): limits the workgroups per CU
template <int TI, int TJ, int SP, int NT, bool NTLOAD>
float timeVariant5(const Shape& s, const char* src, char* dst, const char* order, unsigned KL, int reps) {
  TArgs a{src, dst, s.ei, s.ej, s.ek, s.sj, s.sk, s.di, s.dk, makeWalk(order, (unsigned)(s.ei / TI), (unsigned)(s.ej / TJ), (unsigned)s.ek, KL, 1)};
  const unsigned blocks = (unsigned)(s.ei / TI) * (unsigned)(s.ej / TJ) * (unsigned)s.ek / NT;
  if (g_dyn_lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk5_kernel<TI, TJ, SP, NT, NTLOAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_dyn_lds));
  CK(hipMemset(dst, 0, g_buf_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  walk5_kernel<TI, TJ, SP, NT, NTLOAD><<<blocks, kThreads, g_dyn_lds>>>(a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) walk5_kernel<TI, TJ, SP, NT, NTLOAD><<<blocks, kThreads, g_dyn_lds>>>(a);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / reps;
}
// one cell of the destination per (i, j, k) must equal the source cell: spot check of a phase-5 variant (the harness fills the
// source with a position-dependent pattern for this)
#define ROW5(TI, TJ, SP, NT, NTLOAD, ORDER, KL)                                                                           \
  {                                                                                                                        \
    Shape sf = shapes[0], sb = shapes[1];                                                                                  \
    sf.di += g_far_pad;                                                                                                    \
    sb.sj += g_far_pad;                                                                                                    \
    float f = timeVariant5<TI, TJ, SP, NT, NTLOAD>(sf, src, dst, ORDER, KL, reps);                                          \
    const bool okf = checkMove(sf, src, dst);                                                                              \
    float b = timeVariant5<TI, TJ, SP, NT, NTLOAD>(sb, src, dst, ORDER, KL, reps);                                          \
    const bool okb = checkMove(sb, src, dst);                                                                              \
    printf("| %3d x %3d | store bits %d%s | tiles per wg %d | %-6s KL=%-2d lds+%u far pitch +%lld B | %.3f (%.3f)%s | %.3f (%.3f)%s |\n", TI, TJ, SP, \
           NTLOAD ? "" : ", cached loads", NT, ORDER, KL, g_dyn_lds, g_far_pad * 8, f, g_bytes / f / 8e9, okf ? "" : " WRONG", b,   \
           g_bytes / b / 8e9, okb ? "" : " WRONG");                                                                                           \
    fflush(stdout);                                                                                                        \
  }

static float libLaunch(const Shape& s, char* src, char* dst, int es, int reps, bool sustained, float* per_launch,
                       const KernelTuning* tuning = nullptr) {
  Move3D m;
  m.src_buf = BUF_IN;
  m.dst_buf = BUF_OUT;
  m.extent[0] = s.ei; m.extent[1] = s.ej; m.extent[2] = s.ek;
  m.ss[0] = 1; m.ss[1] = s.sj; m.ss[2] = s.sk;
  m.ds[0] = s.di; m.ds[1] = 1; m.ds[2] = s.dk;
  void* bufs[3] = {src, dst, nullptr};
  std::vector<hipEvent_t> ev(reps + 1);
  for (auto& evt : ev) CK(hipEventCreate(&evt));
  launchMoves(&m, 1, bufs, es, nullptr, tuning);
  CK(hipDeviceSynchronize());
  float total = 0;
  if (sustained) {
    CK(hipEventRecord(ev[0]));
    for (int r = 0; r < reps; ++r) {
      launchMoves(&m, 1, bufs, es, nullptr, tuning);
      CK(hipEventRecord(ev[r + 1]));
    }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
      CK(hipEventElapsedTime(&per_launch[r], ev[r], ev[r + 1]));
      total += per_launch[r];
    }
  } else {
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(ev[0]));
      launchMoves(&m, 1, bufs, es, nullptr, tuning);
      CK(hipEventRecord(ev[1]));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&per_launch[r], ev[0], ev[1]));
      total += per_launch[r];
    }
  }
  for (auto& evt : ev) CK(hipEventDestroy(evt));
  return total / reps;
}

int main(int argc, char** argv) {
  const int es = argc > 1 ? atoi(argv[1]) : 8;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const int phase = argc > 3 ? atoi(argv[3]) : 1;
  // 8-GiB pencils: fp64 / complex64 1024^3, complex128 1024 x 1024 x 512, fp32 2048 x 1024 x 1024 (as bench.py's dtype table)
  const long long NX = es == 4 ? 2048 : 1024, NY = 1024, NZ = es == 16 ? 512 : 1024, E = NX * NY * NZ;
  char *src, *dst;
  g_buf_bytes = (size_t)E * es + (phase == 5 ? (64u << 20) : 0);  // phase 5 pads the far pitch
  CK(hipMalloc(&src, g_buf_bytes));
  CK(hipMalloc(&dst, g_buf_bytes));
  CK(hipMemset(src, 1, E * es));
  if (phase == 5 && es == 8) fill_pattern_k<<<8192, 256>>>(reinterpret_cast<unsigned long long*>(src), (unsigned long long)(g_buf_bytes / 8));
  g_bytes = 2.0 * E * es;
  // fwd (X->Y): source (x, y, z), destination (y, z, x): i = x -> stride NY*NZ out (far), j = y contiguous out, k = z -> stride NY out.
  // bwd (Y->X): source (y, z, x), destination (x, y, z): i = y, j = x (source stride NY*NZ: far), k = z.
  Shape shapes[2] = {{"fwd", NX, NY, NZ, NX, NX * NY, NY * NZ, NY}, {"bwd", NY, NX, NZ, NY * NZ, NY, NX, NX * NY}};
  printf("# element size %d B, %.2f GB per launch, %d launches per figure; ms per launch (fraction of 8 TB/s), fwd | bwd\n", es, g_bytes / 1e9, reps);
  {
    float pl[64];
    for (int rep = 0; rep < 2; ++rep)
      for (auto& s : shapes) {
        const float iso = libLaunch(s, src, dst, es, reps, false, pl);
        float mn = 1e9, mx = 0;
        for (int r = 0; r < reps; ++r) { mn = std::min(mn, pl[r]); mx = std::max(mx, pl[r]); }
        const float sus = libLaunch(s, src, dst, es, reps, true, pl);
        float smn = 1e9, smx = 0;
        for (int r = 1; r < reps; ++r) { smn = std::min(smn, pl[r]); smx = std::max(smx, pl[r]); }
        printf("library %s (%s): isolated %.3f ms (min %.3f max %.3f), sustained %.3f ms (first %.3f, rest min %.3f max %.3f) -> %.1f us per boundary\n",
               s.name, lastKernelName(), iso, mn, mx, sus, pl[0], smn, smx, (sus - iso) * 1e3);
      }
  }
  {  // the four hops of a cycle back to back with ping-pong buffers (what bench.py times), events between the hops
    auto mk = [&](const Shape& sh) {
      Move3D m;
      m.src_buf = BUF_IN; m.dst_buf = BUF_OUT;
      m.extent[0] = sh.ei; m.extent[1] = sh.ej; m.extent[2] = sh.ek;
      m.ss[0] = 1; m.ss[1] = sh.sj; m.ss[2] = sh.sk;
      m.ds[0] = sh.di; m.ds[1] = 1; m.ds[2] = sh.dk;
      return m;
    };
    const Move3D mv[4] = {mk(shapes[0]), mk(shapes[0]), mk(shapes[1]), mk(shapes[1])};
    const int cycles = 6;
    std::vector<hipEvent_t> ev(4 * cycles + 1);
    for (auto& evt : ev) CK(hipEventCreate(&evt));
    for (int pass = 0; pass < 2; ++pass) {
      char *a = src, *b2 = dst;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(ev[0]));
      for (int c = 0; c < cycles; ++c)
        for (int hh = 0; hh < 4; ++hh) {
          void* bufs[3] = {a, b2, nullptr};
          launchMoves(&mv[hh], 1, bufs, es, nullptr);
          CK(hipEventRecord(ev[4 * c + hh + 1]));
          std::swap(a, b2);
        }
      CK(hipDeviceSynchronize());
      float hop[4] = {0, 0, 0, 0}, total = 0;
      for (int c = 1; c < cycles; ++c)
        for (int hh = 0; hh < 4; ++hh) {
          float t;
          CK(hipEventElapsedTime(&t, ev[4 * c + hh], ev[4 * c + hh + 1]));
          hop[hh] += t / (cycles - 1);
          total += t / (cycles - 1);
        }
      printf("ping-pong cycle (fwd fwd bwd bwd, events between hops, %d cycles after one warm-up): hops %.3f %.3f %.3f %.3f ms, cycle %.3f ms\n",
             cycles - 1, hop[0], hop[1], hop[2], hop[3], total);
    }
    for (auto& evt : ev) CK(hipEventDestroy(evt));
  }
  {  // the library with its tile walk forced (tuning switch): 0 = i first, 1 = j first
    float pl[64];
    for (int wo = 0; wo < 2; ++wo) {
      KernelTuning t;
      t.walk_order = wo;
      const float f = libLaunch(shapes[0], src, dst, es, reps, true, pl, &t);
      const float b = libLaunch(shapes[1], src, dst, es, reps, true, pl, &t);
      printf("library with walk_order=%d (%s first): fwd %.3f ms, bwd %.3f ms\n", wo, wo ? "j" : "i", f, b);
    }
  }
  printf("| tile i x j | mode | LDS | walk | fwd ms (frac) | bwd ms (frac) |\n|---|---|---|---|---|---|\n");
  // walks: letters least significant first; i / j = tile index along i / j, k = the batch dim, l / h = low (KL values) and high
  // part of k
  if (es == 8 && phase == 5) {
    // everything with the run walk the library ships for the forward hops ("jlih", 32 planes) and, for the inverse hops' sake,
    // the same variant under "ijk"; pattern data (slower than the constant fill of the earlier phases: compare inside this log)
    for (int rep = 0; rep < 2; ++rep) {
      ROW5(64, 64, 8, 1, true, "jlih", 32)   // = what the library runs for the forward hops
      ROW5(64, 64, 8, 1, true, "ijk", 1)
      // (a) the store's cache bits: 0 none, 1 sc0, 2 sc1, 3 sc0 sc1 (write-through), 4 nt, 5 sc0 nt, 6 sc1 nt, 7 sc0 sc1 nt
      ROW5(64, 64, 0, 1, true, "jlih", 32)
      ROW5(64, 64, 1, 1, true, "jlih", 32)
      ROW5(64, 64, 2, 1, true, "jlih", 32)
      ROW5(64, 64, 3, 1, true, "jlih", 32)
      ROW5(64, 64, 4, 1, true, "jlih", 32)
      ROW5(64, 64, 5, 1, true, "jlih", 32)
      ROW5(64, 64, 6, 1, true, "jlih", 32)
      ROW5(64, 64, 7, 1, true, "jlih", 32)
      // (b) persistent workgroups: 2 / 4 / 8 / 16 consecutive tiles of the walk, the next tile's loads in flight during the stores
      ROW5(64, 64, 8, 2, true, "jlih", 32)
      ROW5(64, 64, 8, 4, true, "jlih", 32)
      ROW5(64, 64, 8, 8, true, "jlih", 32)
      ROW5(64, 64, 8, 16, true, "jlih", 32)
      ROW5(64, 64, 8, 4, true, "ijk", 1)
      ROW5(64, 64, 8, 16, true, "ijk", 1)
      ROW5(64, 128, 8, 4, true, "ijk", 1)
      ROW5(64, 128, 8, 1, true, "ijk", 1)    // = the library's inverse hops
      // (c) fewer workgroups per CU (dynamic LDS nobody uses): 32 KiB tile + 8 / 20 / 48 KiB -> 4 / 3 / 2 per CU instead of 5
      for (unsigned extra : {8u << 10, 20u << 10, 48u << 10}) {
        g_dyn_lds = extra;
        ROW5(64, 64, 8, 1, true, "jlih", 32)
        ROW5(64, 64, 8, 4, true, "jlih", 32)
      }
      g_dyn_lds = 0;
      // (d) shapes with longer destination segments under the run walk
      ROW5(32, 128, 8, 1, true, "jlih", 32)
      ROW5(32, 128, 8, 4, true, "jlih", 32)
      ROW5(16, 256, 8, 1, true, "jlih", 32)
      ROW5(64, 128, 8, 1, true, "jlih", 32)
      ROW5(64, 128, 8, 2, true, "jlih", 32)
      ROW5(128, 32, 8, 4, true, "jlih", 32)
      // (e) DRAM channel aliasing probe: the far-strided rows 8 MiB + 256 B / 4 KiB / 16.25 KiB apart instead of 8 MiB (not a
      // layout the API can produce: a diagnostic of whether the 64 rows of a tile meeting the same channels is what costs)
      for (long long pad : {32ll, 512ll, 2080ll}) {
        g_far_pad = pad;
        ROW5(64, 64, 8, 1, true, "jlih", 32)
        ROW5(64, 64, 8, 1, true, "ijk", 1)
      }
      g_far_pad = 0;
    }
  } else if (es == 8 && phase == 1) {
    for (int rep = 0; rep < 2; ++rep) {
      ROW(8, 2, 64, 64, 2, true, "jik", 1, 1)   // the library's default for fp64
      ROW(8, 2, 64, 64, 2, true, "ijk", 1, 1)
      ROW(8, 2, 64, 64, 2, true, "jik", 1, 0)
      ROW(8, 2, 64, 64, 2, true, "ljih", 8, 1)  // 8 consecutive k planes first: destination rows (i, k..k+7) are 64 KiB runs
      ROW(8, 2, 64, 64, 2, true, "jlih", 8, 1)
      ROW(8, 2, 64, 64, 2, true, "ljih", 4, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 2, true, "kji", 1, 1)
      ROW(8, 2, 64, 64, 2, true, "jki", 1, 1)
      ROW(8, 2, 64, 64, 4, true, "jik", 1, 1)   // cached loads, streaming stores
      ROW(8, 2, 64, 64, 1, true, "jik", 1, 1)   // streaming loads, cached stores
      ROW(8, 2, 64, 64, 0, true, "jik", 1, 1)
      ROW(8, 2, 64, 128, 2, true, "jik", 1, 1)  // 1-KiB destination segments, 64 KiB of LDS
      ROW(8, 2, 64, 128, 2, true, "ijk", 1, 1)
      ROW(8, 2, 32, 128, 2, true, "jik", 1, 1)  // 1-KiB destination segments, 256-byte source segments
      ROW(8, 2, 32, 128, 2, true, "ijk", 1, 1)
      ROW(8, 2, 32, 128, 2, true, "jlih", 8, 1)
      ROW(8, 2, 128, 64, 2, true, "jik", 1, 1)  // 1-KiB source segments
      ROW(8, 2, 128, 64, 2, true, "ijk", 1, 1)
      ROW(8, 2, 128, 32, 2, true, "ijk", 1, 1)
      ROW(8, 2, 32, 64, 2, true, "jik", 1, 1)
      ROW(8, 2, 64, 32, 2, true, "jik", 1, 1)
    }
  } else if (es == 8 && phase == 4) {  // the run walk with other tiles / access modes (the mode A/B of phase 1 used plain j first)
    for (int rep = 0; rep < 2; ++rep) {
      ROW(8, 2, 64, 64, 2, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 4, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 1, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 0, true, "jlih", 32, 1)
      ROW(8, 2, 64, 32, 2, true, "jlih", 32, 1)
      ROW(8, 2, 32, 64, 2, true, "jlih", 32, 1)
      ROW(8, 2, 32, 32, 2, true, "jlih", 32, 1)
      ROW(8, 2, 128, 64, 2, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 2, false, "jlih", 32, 1)
    }
  } else if (es == 8 && phase == 3) {
    for (int rep = 0; rep < 2; ++rep) {
      ROW(8, 2, 64, 64, 2, true, "jik", 1, 1)
      ROW(8, 2, 64, 64, 2, true, "ijk", 1, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 32, 0)
      ROW(8, 2, 64, 64, 2, true, "ajlbh", 2, 1)   // (a / l share KL here: 2 i tiles, then j, then 2 planes ...)
      ROW(8, 2, 64, 64, 2, true, "jalbh", 4, 1)
      ROW(8, 2, 64, 64, 2, true, "jlabh", 4, 1)
      ROW(8, 2, 64, 64, 2, true, "jlabh", 8, 1)
      ROW(8, 2, 64, 64, 2, true, "jlabh", 16, 1)
      ROW(8, 2, 64, 128, 2, true, "ijk", 1, 1)
      ROW(8, 2, 64, 128, 2, true, "iljh", 32, 1)
    }
  } else if (es == 8) {  // phase 2: around the winners of phase 1 (fwd: 64 x 64 j, 32 k planes, i; 128 x 32 -- bwd: 64 x 128)
    for (int rep = 0; rep < 2; ++rep) {
      ROW(8, 2, 64, 64, 2, true, "jik", 1, 1)
      ROW(8, 2, 64, 64, 2, true, "ijk", 1, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 16, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 32, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 64, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 128, 1)
      ROW(8, 2, 64, 64, 2, true, "jlih", 256, 1)
      ROW(8, 2, 64, 64, 2, true, "iljh", 8, 1)   // the mirror for far-strided SOURCE rows: i, then k planes, then j
      ROW(8, 2, 64, 64, 2, true, "iljh", 32, 1)
      ROW(8, 2, 64, 64, 2, true, "iljh", 128, 1)
      ROW(8, 2, 64, 64, 2, true, "ikj", 1, 1)
      ROW(8, 2, 128, 32, 2, true, "ijk", 1, 1)
      ROW(8, 2, 128, 32, 2, true, "jik", 1, 1)
      ROW(8, 2, 128, 32, 2, true, "jlih", 32, 1)
      ROW(8, 2, 128, 32, 2, true, "iljh", 32, 1)
      ROW(8, 2, 256, 16, 2, true, "ijk", 1, 1)
      ROW(8, 2, 64, 128, 2, true, "ijk", 1, 1)
      ROW(8, 2, 64, 128, 2, true, "jik", 1, 1)
      ROW(8, 2, 64, 128, 2, true, "iljh", 32, 1)
      ROW(8, 2, 64, 128, 2, true, "jlih", 32, 1)
      ROW(8, 2, 64, 128, 4, true, "ijk", 1, 1)
      ROW(8, 2, 32, 256, 2, true, "ijk", 1, 1)
      ROW(8, 2, 32, 256, 2, true, "jik", 1, 1)
    }
  } else if (es == 16) {
    for (int rep = 0; rep < 2; ++rep) {
      ROW(16, 1, 32, 32, 2, false, "jik", 1, 1)  // the library's default for 16-byte elements
      ROW(16, 1, 32, 32, 2, false, "ijk", 1, 1)
      ROW(16, 1, 32, 32, 2, false, "jlih", 8, 1)
      ROW(16, 1, 32, 32, 2, false, "jlih", 16, 1)
      ROW(16, 1, 32, 32, 2, false, "jlih", 32, 1)
      ROW(16, 1, 32, 32, 2, false, "jlih", 64, 1)
      ROW(16, 1, 32, 32, 2, false, "iljh", 32, 1)
      ROW(16, 1, 32, 64, 2, false, "jik", 1, 1)  // 1-KiB destination segments
      ROW(16, 1, 32, 64, 2, false, "ijk", 1, 1)
      ROW(16, 1, 32, 64, 2, false, "iljh", 32, 1)
      ROW(16, 1, 64, 32, 2, false, "jik", 1, 1)
      ROW(16, 1, 64, 32, 2, false, "ijk", 1, 1)
      ROW(16, 1, 64, 32, 2, false, "jlih", 32, 1)
      ROW(16, 1, 64, 64, 2, false, "jik", 1, 1)
      ROW(16, 1, 64, 64, 2, false, "ijk", 1, 1)
      ROW(16, 1, 16, 64, 2, false, "jik", 1, 1)
      ROW(16, 1, 64, 16, 2, false, "ijk", 1, 1)
      ROW(16, 1, 64, 16, 2, false, "jlih", 32, 1)
    }
  } else {  // 4-byte elements (library default: 64 x 128 tiles; i first when the far-strided side is the destination)
    for (int rep = 0; rep < 2; ++rep) {
      ROW(4, 4, 64, 128, 2, true, "jik", 1, 1)
      ROW(4, 4, 64, 128, 2, true, "ijk", 1, 1)
      ROW(4, 4, 64, 128, 2, true, "jlih", 32, 1)
      ROW(4, 4, 64, 128, 2, true, "jlih", 8, 1)
      ROW(4, 4, 64, 128, 2, true, "iljh", 32, 1)
      ROW(4, 4, 128, 64, 2, true, "ijk", 1, 1)
      ROW(4, 4, 128, 64, 2, true, "jlih", 32, 1)
      ROW(4, 4, 128, 128, 2, true, "jik", 1, 1)
      ROW(4, 4, 128, 128, 2, true, "ijk", 1, 1)
      ROW(4, 4, 64, 256, 2, true, "ijk", 1, 1)
      ROW(4, 4, 64, 256, 2, true, "jik", 1, 1)
      ROW(4, 4, 256, 64, 2, true, "ijk", 1, 1)
      ROW(4, 4, 256, 32, 2, true, "ijk", 1, 1)
    }
  }
  return 0;
}
