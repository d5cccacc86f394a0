// stride_probe.hip -- where does the far-stride penalty of the permutations come from?  Each 256-thread workgroup
// moves a 64-row x 512-byte tile; the rows of a tile are S bytes apart on the strided side and contiguous (dense tile
// order) on the other side.  Sweeping S separates "many rows per tile" from "rows in distant pages".  Tuning aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// tile t: strided side rows at  base(t) + r * S,  base(t) = (t % (S/512)) * 512 + (t / (S/512)) * 64 * S
template <bool STRIDED_WRITE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, char* __restrict__ dst, size_t S) {
  const size_t t = blockIdx.x;
  const size_t per = S / 512;
  const size_t sbase = (t % per) * 512 + (t / per) * 64 * S;
  const size_t dbase = t * 64 * 512;
  const int lane = threadIdx.x % 32, row0 = threadIdx.x / 32;  // 32 lanes x 16 B = one 512-byte row
  u32x4 v[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = row0 + p * 8;
    const char* a = STRIDED_WRITE ? src + dbase + (size_t)r * 512 : src + sbase + (size_t)r * S;
    v[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a) + lane);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = row0 + p * 8;
    char* a = STRIDED_WRITE ? dst + sbase + (size_t)r * S : dst + dbase + (size_t)r * 512;
    __builtin_nontemporal_store(v[p], reinterpret_cast<u32x4*>(a) + lane);
  }
}

int main() {
  const size_t bytes = (size_t)8 << 30;
  char *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 2, bytes));
  const unsigned tiles = (unsigned)(bytes / (64 * 512));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%12s %16s %16s\n", "row stride", "strided WRITE", "strided READ");
  for (size_t S : {(size_t)512, (size_t)4096, (size_t)8192, (size_t)65536, (size_t)524288, (size_t)1 << 20, (size_t)2 << 20,
                   (size_t)4 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
    float ms[2];
    for (int w = 0; w < 2; ++w) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) {
          if (w == 0) probe<true><<<tiles, 256>>>(a, b, S);
          else probe<false><<<tiles, 256>>>(a, b, S);
        }
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms[w], e0, e1));
      }
      ms[w] /= 5;
    }
    printf("%10zu B %8.3f ms %5.0f GB/s %8.3f ms %5.0f GB/s\n", S, ms[0], 2.0 * bytes / ms[0] / 1e6, ms[1], 2.0 * bytes / ms[1] / 1e6);
  }
  return 0;
}
