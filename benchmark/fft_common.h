// fft_common.h -- pieces shared by the FFT harnesses around the library (benchmark/fft3d_benchmark.cpp,
// examples/cc/poisson.cpp): error-check macros and batched 1-D FFTs along one global axis of a pencil
// (hipFFT/rocFFT; strided plans for non-contiguous axes as in the reference's benchmark.cu:378-411).
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdio>
#include <cstdlib>

#include "cudecomp.h"

#define CHECK_HIP(x)                                                                  \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d HIP error %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)
#define CHECK_FFT(x)                                                                  \
  do {                                                                                \
    hipfftResult r_ = (x);                                                            \
    if (r_ != HIPFFT_SUCCESS) {                                                       \
      fprintf(stderr, "%s:%d hipFFT error %d\n", __FILE__, __LINE__, (int)r_);        \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)
#define CHECK_CD(x)                                                                   \
  do {                                                                                \
    cudecompResult_t r_ = (x);                                                        \
    if (r_ != CUDECOMP_RESULT_SUCCESS) {                                              \
      fprintf(stderr, "%s:%d cuDecomp error %d\n", __FILE__, __LINE__, (int)r_);      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

template <typename C>
__global__ void scale_kernel(C* data, double factor, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    data[i].x *= factor;
    data[i].y *= factor;
  }
}

// 1-D FFTs along global axis `axis` of a pencil, in place
struct AxisFFT {
  hipfftHandle plan = 0;
  int loops = 1;
  long long loop_stride = 0;

  void create(const cudecompPencilInfo_t& p, int axis, bool dbl, hipStream_t stream) {
    int m = 0;
    for (int i = 0; i < 3; ++i)
      if (p.order[i] == axis) m = i;
    int n = p.shape[m];
    const hipfftType type = dbl ? HIPFFT_Z2Z : HIPFFT_C2C;
    if (m == 0) {  // contiguous lines
      CHECK_FFT(hipfftPlanMany(&plan, 1, &n, nullptr, 1, n, nullptr, 1, n, type, p.shape[1] * p.shape[2]));
    } else if (m == 2) {  // lines strided by a whole plane, one batch entry per in-plane point
      const int stride = p.shape[0] * p.shape[1];
      CHECK_FFT(hipfftPlanMany(&plan, 1, &n, &n, stride, 1, &n, stride, 1, type, stride));
    } else {  // middle axis: strided inside a plane, loop over planes (reference benchmark.cu:378-380,528-533)
      const int stride = p.shape[0];
      CHECK_FFT(hipfftPlanMany(&plan, 1, &n, &n, stride, 1, &n, stride, 1, type, stride));
      loops = p.shape[2];
      loop_stride = (long long)p.shape[0] * p.shape[1];
    }
    CHECK_FFT(hipfftSetStream(plan, stream));
  }
  template <typename C>
  void exec(C* data, int direction, bool dbl) {
    for (int l = 0; l < loops; ++l) {
      C* ptr = data + l * loop_stride;
      if (dbl) CHECK_FFT(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)ptr, (hipfftDoubleComplex*)ptr, direction));
      else CHECK_FFT(hipfftExecC2C(plan, (hipfftComplex*)ptr, (hipfftComplex*)ptr, direction));
    }
  }
};

