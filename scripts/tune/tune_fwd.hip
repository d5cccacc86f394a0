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
#define ROW(ES, VW, TI, TJ, STREAM, SWZ, ORDER, KL, XCD)                                                                  \
  {                                                                                                                        \
    float f = timeVariant<ES, VW, TI, TJ, STREAM, SWZ>(shapes[0], src, dst, ORDER, KL, XCD, reps);                          \
    float b = timeVariant<ES, VW, TI, TJ, STREAM, SWZ>(shapes[1], src, dst, ORDER, KL, XCD, reps);                          \
    printf("| %3d x %3d | %d | %-5s | %-6s KL=%-2d xcd=%d | %.3f (%.3f) | %.3f (%.3f) |\n", TI, TJ, STREAM, SWZ ? "swz" : "pad", \
           ORDER, KL, XCD, f, g_bytes / f / 8e9, b, g_bytes / b / 8e9);                                                     \
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
  CK(hipMalloc(&src, E * es));
  CK(hipMalloc(&dst, E * es));
  CK(hipMemset(src, 1, E * es));
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
  if (es == 8 && phase == 1) {
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
