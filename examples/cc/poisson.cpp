// poisson.cpp -- spectral Poisson solver on a pencil-decomposed periodic box: the kind of solver the library
// exists for (counterpart of the reference's examples/cc/poisson).
//
//   laplace(phi) = f on [0, 2 pi)^3, periodic;   f = -(a^2 + b^2 + c^2) sin(a x) cos(b y) sin(c z)
//   => phi = sin(a x) cos(b y) sin(c z)
//
// forward FFT (1-D FFTs along X, transpose X->Y, along Y, transpose Y->Z, along Z), divide every mode by
// -(kx^2 + ky^2 + kz^2) in the Z pencils (global wavenumbers from the pencil's lo / order), inverse FFT back to
// X pencils, compare with the analytic solution.  Double precision; ranks from the launcher environment.
//
//   ./poisson [--n 64] [--pr P --pc Q] [--backend B] [--default-layout]
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <string>
#include <vector>

#include "../../benchmark/fft_common.h"

// phi_hat(k) = f_hat(k) / (-|k|^2) / N, mean mode set to zero; one thread per element of the Z pencil
__global__ void solve_modes(hipfftDoubleComplex* data, cudecompPencilInfo_t p, int gx, int gy, int gz, double norm) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.size) return;
  const int g[3] = {gx, gy, gz};
  long long l[3] = {i % p.shape[0], i / p.shape[0] % p.shape[1], i / ((long long)p.shape[0] * p.shape[1])};
  double k2 = 0;
  for (int m = 0; m < 3; ++m) {
    const int axis = p.order[m];
    long long k = l[m] + p.lo[m];           // global index along that axis
    if (k > g[axis] / 2) k -= g[axis];      // signed wavenumber
    k2 += (double)(k * k);
  }
  const double s = (k2 > 0) ? -norm / k2 : 0.0;
  data[i].x *= s;
  data[i].y *= s;
}

int main(int argc, char** argv) {
  int n = 64, pr = 1, pc = 0, backend = CUDECOMP_TRANSPOSE_COMM_NCCL;
  bool contiguous = true;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--n" && i + 1 < argc) n = atoi(argv[++i]);
    else if (a == "--pr" && i + 1 < argc) pr = atoi(argv[++i]);
    else if (a == "--pc" && i + 1 < argc) pc = atoi(argv[++i]);
    else if (a == "--backend" && i + 1 < argc) backend = atoi(argv[++i]);
    else if (a == "--default-layout") contiguous = false;
  }
  const char* e = getenv("RANK");
  const int rank = e ? atoi(e) : 0;
  e = getenv("WORLD_SIZE");
  const int nranks = e ? atoi(e) : 1;
  if (pc == 0) pc = nranks / pr;
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  e = getenv("LOCAL_RANK");
  CHECK_HIP(hipSetDevice((e ? atoi(e) : rank) % ndev));
  hipStream_t stream = 0;

  cudecompHandle_t handle;
  CHECK_CD(cudecompInit(&handle, MPI_COMM_WORLD));
  cudecompGridDescConfig_t config;
  CHECK_CD(cudecompGridDescConfigSetDefaults(&config));
  for (int i = 0; i < 3; ++i) {
    config.gdims[i] = n;
    config.transpose_axis_contiguous[i] = contiguous;
  }
  config.pdims[0] = pr;
  config.pdims[1] = pc;
  config.transpose_comm_backend = (cudecompTransposeCommBackend_t)backend;
  cudecompGridDesc_t gd;
  CHECK_CD(cudecompGridDescCreate(handle, &gd, &config, nullptr));

  cudecompPencilInfo_t px, py, pz;
  CHECK_CD(cudecompGetPencilInfo(handle, gd, &px, 0, nullptr, nullptr));
  CHECK_CD(cudecompGetPencilInfo(handle, gd, &py, 1, nullptr, nullptr));
  CHECK_CD(cudecompGetPencilInfo(handle, gd, &pz, 2, nullptr, nullptr));
  int64_t ws = 0;
  CHECK_CD(cudecompGetTransposeWorkspaceSize(handle, gd, &ws));
  const int64_t nel = std::max({px.size, py.size, pz.size});
  using C = hipfftDoubleComplex;
  C *data = nullptr, *work = nullptr;
  CHECK_HIP(hipMalloc((void**)&data, nel * sizeof(C)));
  CHECK_CD(cudecompMalloc(handle, gd, (void**)&work, ws * sizeof(C)));

  AxisFFT fx, fy, fz;
  fx.create(px, 0, true, stream);
  fy.create(py, 1, true, stream);
  fz.create(pz, 2, true, stream);

  // right-hand side and analytic solution on this rank's X pencil
  const int ka = 1, kb = 2, kc = 3;
  const double h = 2.0 * M_PI / n;
  std::vector<std::complex<double>> rhs(px.size), got(px.size);
  std::vector<double> exact(px.size);
  for (int64_t i = 0; i < px.size; ++i) {
    int64_t l[3] = {i % px.shape[0], i / px.shape[0] % px.shape[1], i / ((int64_t)px.shape[0] * px.shape[1])};
    double x[3];
    for (int m = 0; m < 3; ++m) x[px.order[m]] = h * (double)(l[m] + px.lo[m]);
    exact[i] = std::sin(ka * x[0]) * std::cos(kb * x[1]) * std::sin(kc * x[2]);
    rhs[i] = -(double)(ka * ka + kb * kb + kc * kc) * exact[i];
  }
  CHECK_HIP(hipMemcpy(data, rhs.data(), px.size * sizeof(C), hipMemcpyHostToDevice));

  fx.exec(data, HIPFFT_FORWARD, true);
  CHECK_CD(cudecompTransposeXToY(handle, gd, data, data, work, CUDECOMP_DOUBLE_COMPLEX, nullptr, nullptr, nullptr, nullptr, stream));
  fy.exec(data, HIPFFT_FORWARD, true);
  CHECK_CD(cudecompTransposeYToZ(handle, gd, data, data, work, CUDECOMP_DOUBLE_COMPLEX, nullptr, nullptr, nullptr, nullptr, stream));
  fz.exec(data, HIPFFT_FORWARD, true);
  solve_modes<<<(unsigned)((pz.size + 255) / 256), 256, 0, stream>>>(data, pz, n, n, n, 1.0 / ((double)n * n * n));
  fz.exec(data, HIPFFT_BACKWARD, true);
  CHECK_CD(cudecompTransposeZToY(handle, gd, data, data, work, CUDECOMP_DOUBLE_COMPLEX, nullptr, nullptr, nullptr, nullptr, stream));
  fy.exec(data, HIPFFT_BACKWARD, true);
  CHECK_CD(cudecompTransposeYToX(handle, gd, data, data, work, CUDECOMP_DOUBLE_COMPLEX, nullptr, nullptr, nullptr, nullptr, stream));
  fx.exec(data, HIPFFT_BACKWARD, true);
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(got.data(), data, px.size * sizeof(C), hipMemcpyDeviceToHost));

  double err = 0;
  for (int64_t i = 0; i < px.size; ++i) err = std::max(err, std::abs(got[i] - std::complex<double>(exact[i], 0.0)));
  const bool ok = err < 1e-11;
  printf("{\"rank\": %d, \"nranks\": %d, \"n\": %d, \"pdims\": [%d, %d], \"max_abs_err\": %.3e, \"ok\": %s}\n", rank, nranks, n,
         config.pdims[0], config.pdims[1], err, ok ? "true" : "false");

  hipfftDestroy(fx.plan);
  hipfftDestroy(fy.plan);
  hipfftDestroy(fz.plan);
  CHECK_CD(cudecompFree(handle, gd, work));
  CHECK_HIP(hipFree(data));
  CHECK_CD(cudecompGridDescDestroy(handle, gd));
  CHECK_CD(cudecompFinalize(handle));
  return ok ? 0 : 1;
}
