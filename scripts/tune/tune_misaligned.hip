// Misaligned (halo-shifted) permutations: A/B of store strategies, fp64, in ONE process.
//   lib      the library's kernel as shipped (cached access, i-first walk for misaligned moves)
//   win<NT>  "line-aligned store windows": the tile of destination row i covers j in [bj*TJ - p_i, +TJ) where p_i is the
//            element phase of that row's start inside a 128-B line, so that every store of the body is a whole, aligned
//            line segment (16-B aligned dwordx4, non-temporal or cached); the load phase fetches the TJ + L - 1 source
//            rows the windows of the tile can touch.
// Shapes: the 1024^3 pencil with a halo of 1 (row pitch 1026) and config 5's 2048-wide pencil with a halo of 2.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e = (x);                                                \
    if (e != hipSuccess) {                                             \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)
using cudecomp::Move3D;
typedef double __attribute__((ext_vector_type(2))) d2;
typedef d2 __attribute__((aligned(8))) d2u;

struct Shape {
  long long ei, ej, ek, sj, sk, di, dk, soff, doff;
  unsigned run = 0;  // walk 5: batch planes per run (all j tiles, `run` planes, then the tile rows i, then the next planes)
};

template <int TI, int TJ, int LDM, int STM, int JFIRST, int L = 16, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void win_kernel(const double* __restrict__ src, double* __restrict__ dst, Shape s,
                                                  unsigned ti_n, unsigned tj_n) {
  // L = elements per alignment unit of the store windows (16 = 128-B line, 8 = 64 B, 4 = 32 B)
  constexpr int ROWS = TJ + L - 1;  // source rows a tile may need
  constexpr int PITCH = TI + 1;
  __shared__ double tile[ROWS * PITCH];
  unsigned lb = blockIdx.x;
  const unsigned nb = gridDim.x, per = nb >> 3;
  if (lb < (per << 3)) lb = (lb & 7u) * per + (lb >> 3);
  unsigned bi, bj, rest;
  if (JFIRST == 1) {
    bj = lb % tj_n; rest = lb / tj_n; bi = rest % ti_n; rest /= ti_n;
  } else if (JFIRST == 0) {
    bi = lb % ti_n; rest = lb / ti_n; bj = rest % tj_n; rest /= tj_n;
  } else if (JFIRST == 5) {  // the library's run walk for far-strided destinations (transpose_kernel, p1 bit 4), over batch planes
    bj = lb % tj_n; rest = lb / tj_n;
    const unsigned klo = rest % s.run; rest /= s.run;
    bi = rest % ti_n; rest = (rest / ti_n) * s.run + klo;
  } else {
    // blocked walk: BI x BJ tiles form a block that is walked first (i fastest inside), then blocks along i, then j
    constexpr unsigned BI = JFIRST == 2 ? 4 : (JFIRST == 3 ? 8 : 2), BJ = JFIRST == 2 ? 4 : (JFIRST == 3 ? 2 : 8);
    const unsigned plane = ti_n * tj_n;
    rest = lb / plane;
    unsigned t = lb % plane;
    const unsigned nbi = ti_n / BI;  // assumes divisibility for the probe shapes (else falls back below)
    if (ti_n % BI == 0 && tj_n % BJ == 0) {
      const unsigned blk = t / (BI * BJ), in = t % (BI * BJ);
      bi = (blk % nbi) * BI + in % BI;
      bj = (blk / nbi) * BJ + in / BI;
    } else {
      bi = t % ti_n; bj = t / ti_n;
    }
  }
  const long long k = rest;
  const long long i0 = (long long)bi * TI, jb = (long long)bj * TJ - (L - 1);
  const double* sp = src + s.soff + k * s.sk;
  double* dp = dst + s.doff + k * s.dk;
  const int tid = threadIdx.x;
  if (LDM == 2) {  // 16-byte ALIGNED loads: shift every row's vectors by its element phase, one scalar load closes the row
    constexpr int TPR = TI / 2, RPP = NTHR / TPR;
    const int l = tid % TPR, lj = tid / TPR;
    const unsigned long long sbase = (unsigned long long)(uintptr_t)sp / 8;
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int jj = lj + p * RPP;
      const long long j = jb + jj;
      if (jj < ROWS && j >= 0 && j < s.ej) {
        const int ph = (int)((sbase + (unsigned long long)(j * s.sj + i0)) & 1);
        const int c0 = 2 * l - ph;  // tile column of the vector's first element
        if (i0 + c0 + 1 < s.ei) {
          const d2 v = *reinterpret_cast<const d2*>(sp + j * s.sj + i0 + c0);
          if (c0 >= 0) tile[jj * PITCH + c0] = v.x;
          tile[jj * PITCH + c0 + 1] = v.y;
        } else if (c0 >= 0 && i0 + c0 < s.ei) {
          tile[jj * PITCH + c0] = sp[j * s.sj + i0 + c0];
        }
        if (ph && l == TPR - 1 && i0 + TI - 1 < s.ei) tile[jj * PITCH + TI - 1] = sp[j * s.sj + i0 + TI - 1];
      }
    }
  } else {  // load: ROWS source rows x TI elements, 2 elements per lane
    constexpr int TPR = TI / 2, RPP = NTHR / TPR;
    const int li = (tid % TPR) * 2, lj = tid / TPR;
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int jj = lj + p * RPP;
      const long long j = jb + jj;
      if (jj < ROWS && j >= 0 && j < s.ej && i0 + li < s.ei) {
        const d2u* q = reinterpret_cast<const d2u*>(sp + j * s.sj + i0 + li);
        d2 v = LDM ? __builtin_nontemporal_load(q) : *q;
        tile[jj * PITCH + li] = v.x;
        tile[jj * PITCH + li + 1] = v.y;
      }
    }
  }
  __syncthreads();
  {  // store: destination row i covers j in [bj*TJ - p_i, +TJ); 2 elements per lane, 16-B aligned
    constexpr int TPO = TJ / 2, RPO = NTHR / TPO;
    const int c = tid % TPO, lr = tid / TPO;
    const unsigned long long dbase = (unsigned long long)(uintptr_t)dp / 8;
#pragma unroll
    for (int p = 0; p < TI / RPO; ++p) {
      const int ii = lr + p * RPO;
      const long long i = i0 + ii;
      if (i >= s.ei) continue;
      const int ph = (int)((dbase + (unsigned long long)(i * s.di)) % L);
      const int r = (L - 1) - ph + 2 * c;  // LDS row of the lane's first element
      const long long j = jb + r;
      double* q = dp + i * s.di + j;
      const double a = tile[r * PITCH + ii], b = tile[(r + 1) * PITCH + ii];
      if (j >= 0 && j + 1 < s.ej) {
        d2 v = {a, b};
        if (STM) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(q));
        else *reinterpret_cast<d2*>(q) = v;
      } else {
        if (j >= 0 && j < s.ej) q[0] = a;
        if (j + 1 >= 0 && j + 1 < s.ej) q[1] = b;
      }
    }
  }
}

// "peel": rectangular tile as in the library; every destination row is written as aligned 16-B granules -- the
// granules of the row's first and last (partial) 128-B line with default caching so that L2 can merge them with the
// neighbouring tile's half, the whole lines in between non-temporally.
template <int TI, int TJ, int LDM, int BODY, bool JFIRST>
__global__ __launch_bounds__(256) void peel_kernel(const double* __restrict__ src, double* __restrict__ dst, Shape s,
                                                   unsigned ti_n, unsigned tj_n) {
  constexpr int L = 16;
  constexpr int PITCH = TI + 1;
  __shared__ double tile[TJ * PITCH];
  unsigned lb = blockIdx.x;
  const unsigned nb = gridDim.x, per = nb >> 3;
  if (lb < (per << 3)) lb = (lb & 7u) * per + (lb >> 3);
  unsigned bi, bj, rest;
  if (JFIRST) {
    bj = lb % tj_n; rest = lb / tj_n; bi = rest % ti_n; rest /= ti_n;
  } else {
    bi = lb % ti_n; rest = lb / ti_n; bj = rest % tj_n; rest /= tj_n;
  }
  const long long k = rest;
  const long long i0 = (long long)bi * TI, j0 = (long long)bj * TJ;
  const double* sp = src + s.soff + k * s.sk;
  double* dp = dst + s.doff + k * s.dk;
  const int tid = threadIdx.x;
  {
    constexpr int TPR = TI / 2, RPP = 256 / TPR;
    const int li = (tid % TPR) * 2, lj = tid / TPR;
#pragma unroll
    for (int p = 0; p < TJ / RPP; ++p) {
      const int jj = lj + p * RPP;
      const long long j = j0 + jj;
      if (j < s.ej && i0 + li < s.ei) {
        const d2u* q = reinterpret_cast<const d2u*>(sp + j * s.sj + i0 + li);
        d2 v = LDM ? __builtin_nontemporal_load(q) : *q;
        tile[jj * PITCH + li] = v.x;
        tile[jj * PITCH + li + 1] = v.y;
      }
    }
  }
  __syncthreads();
  {
    // lanes per destination row: TJ/2 + 1 granules (the row segment may start in the middle of a granule)
    constexpr int TPO = TJ / 2 + 1;
    const unsigned long long dbase = (unsigned long long)(uintptr_t)dp / 8;
    for (int w = tid; w < TI * TPO; w += 256) {
      const int ii = w / TPO, c = w % TPO;
      const long long i = i0 + ii;
      if (i >= s.ei) continue;
      const long long rowstart = (long long)(dbase + (unsigned long long)(i * s.di)) + j0;  // element address of (i, j0)
      const int odd = (int)(rowstart & 1);
      const int r = 2 * c - odd;  // tile column (j - j0) of the granule's first element
      const long long j = j0 + r;
      double* q = dp + i * s.di + j;
      const long long jend = (j0 + TJ < s.ej) ? j0 + TJ : s.ej;
      const bool lo = r >= 0 && j < jend, hi = r + 1 >= 0 && r + 1 < TJ && j + 1 < jend;
      if (lo && hi) {
        d2 v = {tile[r * PITCH + ii], tile[(r + 1) * PITCH + ii]};
        // whole-line body vs the partial first / last line of this row segment
        const long long line = (rowstart + r) / L, first = rowstart / L, last = (rowstart + (jend - j0) - 1) / L;
        const bool partial = (line == first && (rowstart % L) != 0) || (line == last && ((rowstart + (jend - j0)) % L) != 0);
        if (BODY && !partial) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(q));
        else *reinterpret_cast<d2*>(q) = v;
      } else if (lo) {
        q[0] = tile[r * PITCH + ii];
      } else if (hi) {
        q[1] = tile[(r + 1) * PITCH + ii];
      }
    }
  }
}

template <int TI, int TJ, int LDM, int BODY, bool JF>
static void launchPeel(struct Ctx* c);

static float timeIt(void (*fn)(void*), void* ctx, int reps = 8) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  fn(ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) fn(ctx);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Ctx {
  const double* src;
  double* dst;
  Shape s;
  int variant;
  cudecomp::KernelTuning tuning;
  unsigned runs[3] = {1, 1, 1};
};

template <int TI, int TJ, int LDM, int STM, int JF, int L = 16, int NTHR = 256>
static void launchWin(Ctx* c) {
  const unsigned ti = (unsigned)((c->s.ei + TI - 1) / TI), tj = (unsigned)((c->s.ej + L - 1 + TJ - 1) / TJ);
  win_kernel<TI, TJ, LDM, STM, JF, L, NTHR><<<dim3(ti * tj * (unsigned)c->s.ek), NTHR>>>(c->src, c->dst, c->s, ti, tj);
}

template <int TI, int TJ, int LDM, int BODY, bool JF>
static void launchPeel(Ctx* c) {
  const unsigned ti = (unsigned)((c->s.ei + TI - 1) / TI), tj = (unsigned)((c->s.ej + TJ - 1) / TJ);
  peel_kernel<TI, TJ, LDM, BODY, JF><<<dim3(ti * tj * (unsigned)c->s.ek), 256>>>(c->src, c->dst, c->s, ti, tj);
}

static void run(void* p) {
  Ctx* c = (Ctx*)p;
  switch (c->variant) {
    case 0: {
      Move3D m;
      m.src_buf = cudecomp::BUF_IN;
      m.dst_buf = cudecomp::BUF_OUT;
      m.extent[0] = c->s.ei; m.extent[1] = c->s.ej; m.extent[2] = c->s.ek;
      m.ss[0] = 1; m.ss[1] = c->s.sj; m.ss[2] = c->s.sk;
      m.ds[0] = c->s.di; m.ds[1] = 1; m.ds[2] = c->s.dk;
      m.src_off = c->s.soff; m.dst_off = c->s.doff;
      void* bufs[3] = {(void*)c->src, (void*)c->dst, nullptr};
      cudecomp::launchMoves(&m, 1, bufs, 8, nullptr, &c->tuning);
    } break;
    case 1: launchWin<64, 64, 0, 0, 0>(c); break;
    case 2: launchWin<64, 64, 0, 1, 0>(c); break;
    case 3: launchWin<64, 64, 1, 1, 0>(c); break;
    case 4: launchWin<64, 64, 0, 1, 1>(c); break;
    case 5: launchWin<64, 64, 1, 1, 1>(c); break;
    case 6: launchWin<32, 128, 0, 1, 1>(c); break;
    case 7: launchWin<32, 128, 1, 1, 1>(c); break;
    case 8: launchWin<32, 128, 1, 1, 0>(c); break;
    case 9: launchWin<32, 128, 0, 1, 1, 8>(c); break;
    case 10: launchWin<64, 64, 0, 1, 0, 8>(c); break;
    case 11: launchWin<64, 64, 0, 1, 1, 8>(c); break;
    case 12: launchWin<64, 64, 0, 1, 2, 8>(c); break;
    case 13: launchWin<64, 64, 0, 1, 3, 8>(c); break;
    case 14: launchWin<64, 64, 2, 1, 0, 8>(c); break;
    case 15: launchWin<64, 64, 2, 1, 1, 8>(c); break;
    case 16: launchWin<128, 64, 0, 1, 1, 8, 512>(c); break;
    case 17: launchWin<128, 64, 0, 1, 0, 8, 512>(c); break;
    case 18: launchWin<128, 32, 0, 1, 1, 8, 256>(c); break;
    case 19: launchWin<64, 128, 0, 1, 1, 8, 512>(c); break;
    case 20: launchWin<128, 32, 0, 1, 1, 8, 512>(c); break;
    case 21: launchWin<64, 64, 0, 1, 1, 8, 512>(c); break;
    case 22: c->s.run = c->runs[0]; launchWin<64, 64, 0, 1, 5, 8>(c); break;
    case 23: c->s.run = c->runs[1]; launchWin<64, 64, 0, 1, 5, 8>(c); break;
    case 24: c->s.run = c->runs[2]; launchWin<64, 64, 0, 1, 5, 8>(c); break;
  }
}

__global__ void fill(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void diff(const double* a, const double* b, size_t n, unsigned long long* bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (a[i] != b[i]) atomicAdd(bad, 1ull);
}

int main() {
  const size_t n = ((size_t)9 << 30) / 8;
  double *src, *dst, *ref;
  unsigned long long* bad;
  CK(hipMalloc(&src, n * 8));
  CK(hipMalloc(&dst, n * 8));
  CK(hipMalloc(&ref, n * 8));
  CK(hipMalloc(&bad, 8));
  fill<<<4096, 256>>>(src, n);
  struct Named {
    const char* name;
    Shape s;
  };
  auto halo = [](long long X, long long Y, long long Z, long long h, bool fwd) {
    // source pencil (x,y,z) with halo h on every side, destination (y,z,x) [fwd] or (z,x,y) [bwd] with halo h as well
    const long long px = X + 2 * h, py = Y + 2 * h, pz = Z + 2 * h;
    Shape s;
    s.ei = X; s.sj = px; s.soff = h + h * px + h * px * py;
    if (fwd) {  // j = y unit-stride in dst, k = z
      s.ej = Y; s.ek = Z; s.sk = px * py;
      s.di = py * pz; s.dk = py; s.doff = h + h * py + h * py * pz;
    } else {  // bwd: j = z unit-stride in dst, k = y
      s.ej = Z; s.ek = Y; s.sj = px * py; s.sk = px;
      s.di = pz; s.dk = pz * px; s.doff = h + h * pz + h * pz * px;
    }
    return s;
  };
  Named cases[] = {{"1024^3 halo 1 fwd", halo(1024, 1024, 1022, 1, true)},
                   {"1024^3 halo 1 bwd", halo(1024, 1022, 1024, 1, false)},
                   {"2048x1024x256 halo 2 fwd (config 5 pencil)", halo(2048, 1024, 256, 2, true)},
                   {"1024^3 aligned fwd", halo(1024, 1024, 1024, 0, true)},
                   {"1024^3 aligned bwd", halo(1024, 1024, 1024, 0, false)}};
  const char* vn[] = {"lib (cached, auto walk)", "win 64x64 cached/cached i-first", "win 64x64 cached/NT i-first",
                      "win 64x64 NT/NT i-first", "win 64x64 cached/NT j-first", "win 64x64 NT/NT j-first",
                      "win 32x128 cached/NT j-first", "win 32x128 NT/NT j-first", "win 32x128 NT/NT i-first", "win 32x128 c/NT j-first, 64-B units",
                      "win 64x64 c/NT 64B walk i-first", "win 64x64 c/NT 64B walk j-first", "win 64x64 c/NT 64B walk 4x4",
                      "win 64x64 c/NT 64B walk 8ix2j", "win 64x64 ALIGNED-LOADS/NT 64B i-first", "win 64x64 ALIGNED-LOADS/NT 64B j-first",
                      "win 128x64 512thr c/NT 64B j-first", "win 128x64 512thr c/NT 64B i-first", "win 128x32 256thr c/NT 64B j-first",
                      "win 64x128 512thr c/NT 64B j-first", "win 128x32 512thr c/NT 64B j-first", "win 64x64 512thr c/NT 64B j-first",
                      "win 64x64 c/NT 64B RUNS over planes (a)", "win 64x64 c/NT 64B RUNS over planes (b)", "win 64x64 c/NT 64B RUNS over planes (c)"};
  for (auto& c : cases) {
    const double bytes = 2.0 * c.s.ei * c.s.ej * c.s.ek * 8;
    printf("== %s: %lld x %lld x %lld, %.2f GB per launch\n", c.name, c.s.ei, c.s.ej, c.s.ek, bytes / 1e9);
    Ctx ctx{src, ref, c.s, 0, {}};
    CK(hipMemset(ref, 0, n * 8));
    run(&ctx);  // reference result from the library kernel
    CK(hipDeviceSynchronize());
    // divisors of the plane count near 8 / 32 / 128 planes per run (64 KiB / 256 KiB / 1 MiB of every destination row)
    unsigned runs[3] = {1, 1, 1};
    {
      const unsigned want[3] = {8, 32, 128};
      for (int t = 0; t < 3; ++t) {
        unsigned best = 1;
        for (unsigned d = 1; d <= (unsigned)c.s.ek; ++d)
          if (c.s.ek % d == 0 && (best == 1 || (d > best ? d - want[t] : want[t] - d) < (best > want[t] ? best - want[t] : want[t] - best)) && d <= 4 * want[t]) best = d;
        runs[t] = best;
      }
      printf("  (runs over planes: %u / %u / %u)\n", runs[0], runs[1], runs[2]);
    }
    for (int v = (getenv("TUNE_FROM") ? atoi(getenv("TUNE_FROM")) : 0); v < 25; ++v) {
      if (getenv("TUNE_FROM") && v != 0 && v != 10 && v != 11 && v < 22) continue;
      Ctx x{src, dst, c.s, v, {}};
      x.runs[0] = runs[0]; x.runs[1] = runs[1]; x.runs[2] = runs[2];
      CK(hipMemset(dst, 0, n * 8));
      const float ms = timeIt(run, &x);
      CK(hipMemset(bad, 0, 8));
      diff<<<4096, 256>>>(dst, ref, n, bad);
      unsigned long long hb = 0;
      CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      printf("  %-36s %7.3f ms %6.0f GB/s  %s\n", vn[v], ms, bytes / ms / 1e6, hb ? "WRONG" : "ok");
    }
    // library with the other streaming modes for reference
    for (int mode = 1; mode <= 2; ++mode) {
      Ctx x{src, dst, c.s, 0, {}};
      (void)mode;  // (the store-mode override of round 2 is gone: misaligned destinations use cached stores or the window kernel)
      x.tuning.force_streaming = true;
      const float ms = timeIt(run, &x);
      printf("  lib, streaming mode %d                %7.3f ms %6.0f GB/s\n", mode, ms, bytes / ms / 1e6);
    }
    {  // the library's window kernel with 128 x 64 tiles and 512 threads (CUDECOMP_WINDOW_WIDE=1)
      Ctx x{src, dst, c.s, 0, {}};
      x.tuning.window_wide = 1;
      CK(hipMemset(dst, 0, n * 8));
      const float ms = timeIt(run, &x);
      CK(hipMemset(bad, 0, 8));
      diff<<<4096, 256>>>(dst, ref, n, bad);
      unsigned long long hb = 0;
      CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
      printf("  lib, wide window tiles (128x64, 512)  %7.3f ms %6.0f GB/s  %s\n", ms, bytes / ms / 1e6, hb ? "WRONG" : "ok");
    }
  }
  return 0;
}
