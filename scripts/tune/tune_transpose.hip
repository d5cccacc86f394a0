// Standalone tuning harness for the LDS-tiled transpose kernel (not part of the product).
// Times kernel variants on the four hop shapes of the 1024^3 fp64 axis-contiguous cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const char* src; char* dst;
  long long ei, ej, ek, sj, sk, di, dk;
  unsigned ti_n, tj_n;
};

// ORDER: 0 = i fastest, 1 = j fastest, 2 = k fastest (batch), 3 = 8x8 supertiles (i fastest inside)
template <int ORDER>
__device__ __forceinline__ void decode(const Args& a, unsigned lb, unsigned& bi, unsigned& bj, long long& k) {
  if (ORDER == 0) { bi = lb % a.ti_n; unsigned r = lb / a.ti_n; bj = r % a.tj_n; k = r / a.tj_n; }
  else if (ORDER == 1) { bj = lb % a.tj_n; unsigned r = lb / a.tj_n; bi = r % a.ti_n; k = r / a.ti_n; }
  else if (ORDER == 2) { k = lb % (unsigned)a.ek; unsigned r = lb / (unsigned)a.ek; bi = r % a.ti_n; bj = r / a.ti_n; }
  else {
    const unsigned G = 8;
    unsigned per_k = a.ti_n * a.tj_n; k = lb / per_k; unsigned r = lb % per_k;
    unsigned gi_n = a.ti_n / G; unsigned grp = r / (G * G), in = r % (G * G);
    bi = (grp % gi_n) * G + in % G; bj = (grp / gi_n) * G + in / G;
  }
}

template <int TI, int TJ, int NT, int ORDER, bool NTS, bool NTL>
__global__ __launch_bounds__(NT) void tk(const Args a) {
  constexpr int VW = 2;
  constexpr int TPR = TI / VW, RPP = NT / TPR, NP = TJ / RPP;
  constexpr int TPO = TJ / VW, RPO = NT / TPO, NPO = TI / RPO;
  constexpr int PITCH = TI + 1;
  __shared__ u32x2 tile[TJ * PITCH];
  unsigned bi, bj; long long k;
  decode<ORDER>(a, blockIdx.x, bi, bj, k);
  const long long i0 = (long long)bi * TI, j0 = (long long)bj * TJ;
  const u32x2* __restrict__ src = reinterpret_cast<const u32x2*>(a.src) + k * a.sk;
  u32x2* __restrict__ dst = reinterpret_cast<u32x2*>(a.dst) + k * a.dk;
  const int tid = threadIdx.x;
  {
    const int li = (tid % TPR) * VW, lj = tid / TPR;
    u32x4 regs[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const long long j = j0 + lj + p * RPP;
      const u32x4* ptr = reinterpret_cast<const u32x4*>(src + j * a.sj + i0 + li);
      regs[p] = NTL ? __builtin_nontemporal_load(ptr) : *ptr;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      u32x2* row = tile + (lj + p * RPP) * PITCH + li;
      row[0] = regs[p].xy; row[1] = regs[p].zw;
    }
  }
  __syncthreads();
  {
    const int lj = (tid % TPO) * VW, li = tid / TPO;
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
      const int ii = li + p * RPO;
      u32x4 out;
      out.xy = tile[(lj + 0) * PITCH + ii];
      out.zw = tile[(lj + 1) * PITCH + ii];
      u32x4* ptr = reinterpret_cast<u32x4*>(dst + (i0 + ii) * a.di + j0 + lj);
      if (NTS) __builtin_nontemporal_store(out, ptr); else *ptr = out;
    }
  }
}

// variant 2: XCD-aware block remap + MT consecutive tiles (in ORDER sequence) per block
template <int TI, int TJ, int NT, int ORDER, bool XCD, int MT>
__global__ __launch_bounds__(NT) void tk2(const Args a, unsigned nblocks_logical) {
  constexpr int VW = 2;
  constexpr int TPR = TI / VW, RPP = NT / TPR, NP = TJ / RPP;
  constexpr int TPO = TJ / VW, RPO = NT / TPO, NPO = TI / RPO;
  constexpr int PITCH = TI + 1;
  __shared__ u32x2 tile[TJ * PITCH];
  unsigned b = blockIdx.x;
  if (XCD) { unsigned per = gridDim.x / 8; b = (b % 8) * per + b / 8; }
  const int tid = threadIdx.x;
  for (int t = 0; t < MT; ++t) {
    unsigned bi, bj; long long k;
    decode<ORDER>(a, b * MT + t, bi, bj, k);
    const long long i0 = (long long)bi * TI, j0 = (long long)bj * TJ;
    const u32x2* __restrict__ src = reinterpret_cast<const u32x2*>(a.src) + k * a.sk;
    u32x2* __restrict__ dst = reinterpret_cast<u32x2*>(a.dst) + k * a.dk;
    {
      const int li = (tid % TPR) * VW, lj = tid / TPR;
      u32x4 regs[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const long long j = j0 + lj + p * RPP;
        regs[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + j * a.sj + i0 + li));
      }
      if (t > 0) __syncthreads();
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        u32x2* row = tile + (lj + p * RPP) * PITCH + li;
        row[0] = regs[p].xy; row[1] = regs[p].zw;
      }
    }
    __syncthreads();
    {
      const int lj = (tid % TPO) * VW, li = tid / TPO;
#pragma unroll
      for (int p = 0; p < NPO; ++p) {
        const int ii = li + p * RPO;
        u32x4 out;
        out.xy = tile[(lj + 0) * PITCH + ii];
        out.zw = tile[(lj + 1) * PITCH + ii];
        __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(dst + (i0 + ii) * a.di + j0 + lj));
      }
    }
  }
}

struct Shape { const char* name; long long ei, ej, ek, sj, sk, di, dk; };

template <int TI, int TJ, int NT, int ORDER, bool NTS, bool NTL>
float run(const Shape& s, const char* src, char* dst, int reps) {
  Args a{src, dst, s.ei, s.ej, s.ek, s.sj, s.sk, s.di, s.dk, (unsigned)(s.ei / TI), (unsigned)(s.ej / TJ)};
  unsigned blocks = a.ti_n * a.tj_n * (unsigned)s.ek;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  tk<TI, TJ, NT, ORDER, NTS, NTL><<<blocks, NT>>>(a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) tk<TI, TJ, NT, ORDER, NTS, NTL><<<blocks, NT>>>(a);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <bool NTS, bool NTL>
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ s, u32x4* __restrict__ d, long long n) {
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = NTL ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
  for (int u = 0; u < 4; ++u) { if (NTS) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
}
template <bool NTS, bool NTL>
void runcopy(const char* src, char* dst, long long bytes) {
  long long n = bytes / 16;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  copyk<NTS, NTL><<<n / 1024, 256>>>((const u32x4*)src, (u32x4*)dst, n);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) copyk<NTS, NTL><<<n / 1024, 256>>>((const u32x4*)src, (u32x4*)dst, n);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("copy NTS=%d NTL=%d : %.3f ms %.0f GB/s\n", NTS, NTL, ms / 10, 2.0 * bytes / (ms / 10) / 1e6);
}

template <int TI, int TJ, int NT, int ORDER, bool XCD, int MT>
float run2(const Shape& s, const char* src, char* dst, int reps) {
  Args a{src, dst, s.ei, s.ej, s.ek, s.sj, s.sk, s.di, s.dk, (unsigned)(s.ei / TI), (unsigned)(s.ej / TJ)};
  unsigned blocks = a.ti_n * a.tj_n * (unsigned)s.ek / MT;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  tk2<TI, TJ, NT, ORDER, XCD, MT><<<blocks, NT>>>(a, blocks);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) tk2<TI, TJ, NT, ORDER, XCD, MT><<<blocks, NT>>>(a, blocks);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

float runlib(const Shape& s, char* src, char* dst, int reps) {
  cudecomp::Move3D m;
  m.src_buf = cudecomp::BUF_IN; m.dst_buf = cudecomp::BUF_OUT;
  m.extent[0] = s.ei; m.extent[1] = s.ej; m.extent[2] = s.ek;
  m.ss[0] = 1; m.ss[1] = s.sj; m.ss[2] = s.sk;
  m.ds[0] = s.di; m.ds[1] = 1; m.ds[2] = s.dk;
  void* bufs[3] = {src, dst, nullptr};
  fprintf(stderr, "runlib %s\n", s.name);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  cudecomp::launchMoves(&m, 1, bufs, 8, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) cudecomp::launchMoves(&m, 1, bufs, 8, nullptr);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const long long N = 1024, E = N * N * N;
  char *src, *dst;
  CK(hipMalloc(&src, E * 8)); CK(hipMalloc(&dst, E * 8));
  CK(hipMemset(src, 1, E * 8));
  // XToY / YToZ (same structure): i -> stride N*N out, j contiguous out, k -> stride N out
  // ZToY / YToX: i -> stride N out (di = N), j (stride N*N in) contiguous out, k (stride N in) -> stride N*N out
  Shape shapes[2] = {{"fwd (XtoY,YtoZ)", N, N, N, N, N * N, N * N, N},
                     {"bwd (ZtoY,YtoX)", N, N, N, N * N, N, N, N * N}};
  const double bytes = 2.0 * E * 8;
  runcopy<false, false>(src, dst, E * 8); runcopy<true, false>(src, dst, E * 8); runcopy<true, true>(src, dst, E * 8); runcopy<false, true>(src, dst, E * 8);
#define RUN(TI, TJ, NT, ORDER, NTS, NTL)                                                          \
  for (auto& s : shapes) {                                                                          \
    float ms = run<TI, TJ, NT, ORDER, NTS, NTL>(s, src, dst, 10);                                   \
    printf("%-16s TI=%3d TJ=%3d NT=%3d ORDER=%d NTS=%d NTL=%d : %.3f ms  %.0f GB/s\n", s.name, TI, TJ, NT, ORDER, NTS, \
           NTL, ms, bytes / ms / 1e6);                                                              \
  }
#define RUN2(TI, TJ, NT, ORDER, XCD, MT)                                                         \
  for (auto& s : shapes) {                                                                          \
    float ms = run2<TI, TJ, NT, ORDER, XCD, MT>(s, src, dst, 10);                                   \
    printf("%-16s v2 TI=%3d TJ=%3d NT=%3d ORDER=%d XCD=%d MT=%d : %.3f ms  %.0f GB/s\n", s.name, TI, TJ, NT, ORDER, XCD, \
           MT, ms, bytes / ms / 1e6);                                                              \
  }
  setvbuf(stdout, nullptr, _IONBF, 0);
  fprintf(stderr, "entering library loop\n");
  for (int rep = 0; rep < 2; ++rep) {
  for (auto& s : shapes) { float ms = runlib(s, src, dst, 10); printf("%-16s LIBRARY kernel : %.3f ms %.0f GB/s\n", s.name, ms, bytes / ms / 1e6); }
  RUN2(64, 64, 256, 0, true, 1)
  RUN2(64, 64, 256, 1, true, 1)
  RUN2(32, 128, 256, 0, true, 1)
  RUN2(32, 128, 256, 1, true, 1)
  RUN2(16, 256, 256, 0, true, 1)
  RUN2(16, 256, 256, 1, true, 1)
  RUN2(16, 256, 256, 1, false, 1)
  RUN2(8, 512, 256, 1, true, 1)
  RUN2(8, 512, 256, 1, false, 1)
  RUN2(128, 32, 256, 0, true, 1)
  RUN2(256, 16, 256, 0, true, 1)
  RUN2(16, 128, 256, 1, true, 1)
  RUN2(16, 128, 256, 1, false, 1)
  }
  return 0;
}
