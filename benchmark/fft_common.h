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

template <typename R>
__global__ void scale_real_kernel(R* data, R factor, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) data[i] *= factor;
}

// FFT over the `rank` FASTEST memory dims of a pencil at once (they must be complete on this rank), batched over the
// remaining dims: the slab shortcuts of the reference's benchmark (benchmark.cu:340-373: one 2-D FFT instead of two 1-D
// passes and a transpose when a process-grid dim is 1, one 3-D FFT on a single rank).  `real_x` > 0 selects the
// real-to-complex flavour (benchmark.cu:238-330): the pencil then is the COMPLEX X pencil, real_x/2+1 long along x (which
// must be its fastest dim), and the same memory viewed as rows of 2*(real_x/2+1) reals holds the real field in place.
struct BlockFFT {
  hipfftHandle fwd = 0, inv = 0;
  bool real = false, dbl = false;

  void create(const cudecompPencilInfo_t& p, int rank, bool dbl_, int real_x, hipStream_t stream) {
    dbl = dbl_;
    real = real_x > 0;
    int n[3], cdim[3], rdim[3];  // slowest first, as hipFFT wants them
    long long cdist = 1, batch = 1;
    for (int i = 0; i < rank; ++i) {
      const int m = rank - 1 - i;  // memory dim
      n[i] = cdim[i] = rdim[i] = p.shape[m];
      cdist *= p.shape[m];
    }
    for (int m = rank; m < 3; ++m) batch *= p.shape[m];
    if (!real) {
      const hipfftType type = dbl ? HIPFFT_Z2Z : HIPFFT_C2C;
      CHECK_FFT(hipfftPlanMany(&fwd, rank, n, nullptr, 1, (int)cdist, nullptr, 1, (int)cdist, type, (int)batch));
      inv = fwd;
    } else {
      n[rank - 1] = real_x;               // logical length along x
      rdim[rank - 1] = 2 * p.shape[0];    // reals per row (padded), complex per row = p.shape[0] = real_x/2+1
      CHECK_FFT(hipfftPlanMany(&fwd, rank, n, rdim, 1, (int)(2 * cdist), cdim, 1, (int)cdist, dbl ? HIPFFT_D2Z : HIPFFT_R2C, (int)batch));
      CHECK_FFT(hipfftPlanMany(&inv, rank, n, cdim, 1, (int)cdist, rdim, 1, (int)(2 * cdist), dbl ? HIPFFT_Z2D : HIPFFT_C2R, (int)batch));
      CHECK_FFT(hipfftSetStream(inv, stream));
    }
    CHECK_FFT(hipfftSetStream(fwd, stream));
  }
  template <typename C>
  void exec(C* data, int direction) {
    if (!real) {
      if (dbl) CHECK_FFT(hipfftExecZ2Z(fwd, (hipfftDoubleComplex*)data, (hipfftDoubleComplex*)data, direction));
      else CHECK_FFT(hipfftExecC2C(fwd, (hipfftComplex*)data, (hipfftComplex*)data, direction));
    } else if (direction == HIPFFT_FORWARD) {
      if (dbl) CHECK_FFT(hipfftExecD2Z(fwd, (hipfftDoubleReal*)data, (hipfftDoubleComplex*)data));
      else CHECK_FFT(hipfftExecR2C(fwd, (hipfftReal*)data, (hipfftComplex*)data));
    } else {
      if (dbl) CHECK_FFT(hipfftExecZ2D(inv, (hipfftDoubleComplex*)data, (hipfftDoubleReal*)data));
      else CHECK_FFT(hipfftExecC2R(inv, (hipfftComplex*)data, (hipfftReal*)data));
    }
  }
  void destroy() {
    if (inv && inv != fwd) hipfftDestroy(inv);
    if (fwd) hipfftDestroy(fwd);
    fwd = inv = 0;
  }
};

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

