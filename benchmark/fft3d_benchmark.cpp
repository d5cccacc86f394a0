// fft3d_benchmark.cpp -- distributed 3-D complex-to-complex FFT on pencil-decomposed data: batched 1-D
// FFTs with hipFFT/rocFFT on the local pencil + the four cuDecomp transposes, forward then inverse.
//
// This is the canonical CALLER of the transpose path and the counterpart of the reference's
// benchmark/benchmark.cu (C2C flavour; flow :489-611, strided plans for non-contiguous axes :378-411, FLOP
// model and timing protocol :499-505,587-590,658, tolerances :23-27).  BASELINE.json config 4.  It is a
// harness around the library, not part of it.
//
//   ./fft3d_benchmark --gx 256 --gy 256 --gz 256 [--pr P --pc Q] [--backend B] [--double]
//                     [--default-layout] [-o] [--warmup W] [--trials T] [--no-spectrum-check]
//
// One process per rank; ranks are discovered by the library from RANK/WORLD_SIZE (or PMI_*/OMPI_*) like in
// examples/c/basic_usage.c.  Every rank prints one JSON line; benchmark/run_fft3d.py launches the ranks
// and reduces them.
//
// Checks (both must pass):
//   1. spectrum: the forward transform of a plane wave exp(2*pi*i*(kx x/X + ky y/Y + kz z/Z)) is X*Y*Z at
//      (kx,ky,kz) and 0 elsewhere -- verified on the distributed Z pencils through their global indices;
//   2. round trip: forward + inverse + 1/N scaling reproduces uniform random input within 5e-4 (single) /
//      1e-10 (double) max-abs, the reference's tolerances.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "cudecomp.h"

#include "fft_common.h"

struct Options {
  int g[3] = {256, 256, 256};
  int pr = 0, pc = 0, backend = 0;
  bool dbl = false, contiguous = true, out_of_place = false, spectrum = true;
  int warmup = 3, trials = 5;
};

template <typename Real>
int run(const Options& o, int rank, int nranks) {
  using C = typename std::conditional<std::is_same<Real, double>::value, hipfftDoubleComplex, hipfftComplex>::type;
  const cudecompDataType_t dtype = o.dbl ? CUDECOMP_DOUBLE_COMPLEX : CUDECOMP_FLOAT_COMPLEX;
  const double tol = o.dbl ? 1e-10 : 5e-4;
  hipStream_t stream = 0;

  cudecompHandle_t handle;
  CHECK_CD(cudecompInit(&handle, MPI_COMM_WORLD));
  cudecompGridDescConfig_t config;
  CHECK_CD(cudecompGridDescConfigSetDefaults(&config));
  for (int i = 0; i < 3; ++i) {
    config.gdims[i] = o.g[i];
    config.transpose_axis_contiguous[i] = o.contiguous;
  }
  config.pdims[0] = o.pr;
  config.pdims[1] = o.pc;
  if (o.backend) config.transpose_comm_backend = (cudecompTransposeCommBackend_t)o.backend;
  cudecompGridDescAutotuneOptions_t options;
  CHECK_CD(cudecompGridDescAutotuneOptionsSetDefaults(&options));
  options.dtype = dtype;
  options.autotune_transpose_backend = (o.backend == 0);
  for (int i = 0; i < 4; ++i) options.transpose_use_inplace_buffers[i] = !o.out_of_place;
  cudecompGridDesc_t gd;
  const bool tune = (o.backend == 0) || (o.pr == 0 && o.pc == 0);
  if (rank != 0 || tune) fflush(stdout);
  CHECK_CD(cudecompGridDescCreate(handle, &gd, &config, tune ? &options : nullptr));

  cudecompPencilInfo_t px, py, pz;
  CHECK_CD(cudecompGetPencilInfo(handle, gd, &px, 0, nullptr, nullptr));
  CHECK_CD(cudecompGetPencilInfo(handle, gd, &py, 1, nullptr, nullptr));
  CHECK_CD(cudecompGetPencilInfo(handle, gd, &pz, 2, nullptr, nullptr));
  int64_t ws = 0;
  CHECK_CD(cudecompGetTransposeWorkspaceSize(handle, gd, &ws));
  const int64_t nel = std::max({px.size, py.size, pz.size});

  C *data = nullptr, *data2 = nullptr, *work = nullptr;
  CHECK_HIP(hipMalloc((void**)&data, nel * sizeof(C)));
  if (o.out_of_place) CHECK_HIP(hipMalloc((void**)&data2, nel * sizeof(C)));
  CHECK_CD(cudecompMalloc(handle, gd, (void**)&work, ws * sizeof(C)));

  AxisFFT fx, fy, fz;
  fx.create(px, 0, o.dbl, stream);
  fy.create(py, 1, o.dbl, stream);
  fz.create(pz, 2, o.dbl, stream);

  C* in = data;
  C* out = o.out_of_place ? data2 : data;
  auto forward = [&]() {
    fx.exec(in, HIPFFT_FORWARD, o.dbl);
    CHECK_CD(cudecompTransposeXToY(handle, gd, in, out, work, dtype, nullptr, nullptr, nullptr, nullptr, stream));
    fy.exec(out, HIPFFT_FORWARD, o.dbl);
    std::swap(in, out);
    CHECK_CD(cudecompTransposeYToZ(handle, gd, in, out, work, dtype, nullptr, nullptr, nullptr, nullptr, stream));
    fz.exec(out, HIPFFT_FORWARD, o.dbl);
    std::swap(in, out);  // result is in `in`
  };
  auto inverse = [&]() {
    fz.exec(in, HIPFFT_BACKWARD, o.dbl);
    CHECK_CD(cudecompTransposeZToY(handle, gd, in, out, work, dtype, nullptr, nullptr, nullptr, nullptr, stream));
    fy.exec(out, HIPFFT_BACKWARD, o.dbl);
    std::swap(in, out);
    CHECK_CD(cudecompTransposeYToX(handle, gd, in, out, work, dtype, nullptr, nullptr, nullptr, nullptr, stream));
    fx.exec(out, HIPFFT_BACKWARD, o.dbl);
    std::swap(in, out);
  };
  if (!o.out_of_place) out = in = data;

  const double N = (double)o.g[0] * o.g[1] * o.g[2];
  std::vector<std::complex<Real>> host(nel);

  // ---- check 1: spectrum of a plane wave -----------------------------------------------------------
  double spec_err = 0;
  if (o.spectrum) {
    const int k[3] = {3 % o.g[0], 5 % o.g[1], 7 % o.g[2]};
    for (int64_t i = 0; i < px.size; ++i) {
      int64_t l[3] = {i % px.shape[0], i / px.shape[0] % px.shape[1], i / ((int64_t)px.shape[0] * px.shape[1])};
      double phase = 0;
      for (int m = 0; m < 3; ++m) phase += 2.0 * M_PI * k[px.order[m]] * (double)(l[m] + px.lo[m]) / o.g[px.order[m]];
      host[i] = std::complex<Real>((Real)std::cos(phase), (Real)std::sin(phase));
    }
    in = data;
    out = o.out_of_place ? data2 : data;
    CHECK_HIP(hipMemcpy(in, host.data(), px.size * sizeof(C), hipMemcpyHostToDevice));
    forward();
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(host.data(), in, pz.size * sizeof(C), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < pz.size; ++i) {
      int64_t l[3] = {i % pz.shape[0], i / pz.shape[0] % pz.shape[1], i / ((int64_t)pz.shape[0] * pz.shape[1])};
      bool peak = true;
      for (int m = 0; m < 3; ++m) peak = peak && (l[m] + pz.lo[m] == k[pz.order[m]]);
      const std::complex<double> expect(peak ? N : 0.0, 0.0);
      spec_err = std::max(spec_err, std::abs(std::complex<double>(host[i]) - expect) / N);
    }
  }

  // ---- check 2 + timing: random data, forward + inverse -------------------------------------------------
  std::vector<std::complex<Real>> ref(px.size);
  std::default_random_engine rng(1234 + rank);
  std::uniform_real_distribution<Real> dist(0, 1);
  for (auto& v : ref) v = std::complex<Real>(dist(rng), dist(rng));
  in = data;
  out = o.out_of_place ? data2 : data;
  CHECK_HIP(hipMemcpy(in, ref.data(), px.size * sizeof(C), hipMemcpyHostToDevice));

  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  std::vector<double> times;
  for (int t = 0; t < o.warmup + o.trials; ++t) {
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipEventRecord(e0, stream));
    forward();
    inverse();
    CHECK_HIP(hipEventRecord(e1, stream));
    CHECK_HIP(hipDeviceSynchronize());
    float ms = 0;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (t >= o.warmup) times.push_back(ms / 2);  // time of ONE direction, as the reference reports
    scale_kernel<<<(unsigned)((px.size + 255) / 256), 256, 0, stream>>>(in, 1.0 / N, (long long)px.size);  // untimed
  }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(host.data(), in, px.size * sizeof(C), hipMemcpyDeviceToHost));
  double rt_err = 0;
  for (int64_t i = 0; i < px.size; ++i) rt_err = std::max<double>(rt_err, std::abs(host[i] - ref[i]));

  std::sort(times.begin(), times.end());
  double avg = 0;
  for (double t : times) avg += t;
  avg /= times.size();
  const double gflop = 5.0 * N * std::log2(N) * 1e-9;
  const bool ok = rt_err <= tol && (!o.spectrum || spec_err <= (o.dbl ? 1e-12 : 1e-4));
  printf("{\"rank\": %d, \"nranks\": %d, \"gdims\": [%d, %d, %d], \"pdims\": [%d, %d], \"backend\": \"%s\", "
         "\"dtype\": \"%s\", \"layout\": \"%s\", \"out_of_place\": %s, \"ms_min\": %.4f, \"ms_avg\": %.4f, "
         "\"ms_max\": %.4f, \"gflops\": %.1f, \"roundtrip_max_abs_err\": %.3e, \"spectrum_rel_err\": %.3e, "
         "\"tolerance\": %.1e, \"ok\": %s}\n",
         rank, nranks, o.g[0], o.g[1], o.g[2], config.pdims[0], config.pdims[1],
         cudecompTransposeCommBackendToString(config.transpose_comm_backend), o.dbl ? "c128" : "c64",
         o.contiguous ? "axis-contiguous" : "default", o.out_of_place ? "true" : "false", times.front(), avg,
         times.back(), gflop / (avg * 1e-3), rt_err, spec_err, tol, ok ? "true" : "false");
  fflush(stdout);

  hipfftDestroy(fx.plan);
  hipfftDestroy(fy.plan);
  hipfftDestroy(fz.plan);
  CHECK_CD(cudecompFree(handle, gd, work));
  CHECK_HIP(hipFree(data));
  if (data2) CHECK_HIP(hipFree(data2));
  CHECK_CD(cudecompGridDescDestroy(handle, gd));
  CHECK_CD(cudecompFinalize(handle));
  return ok ? 0 : 1;
}

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() { return (i + 1 < argc) ? atoi(argv[++i]) : 0; };
    if (a == "--gx") o.g[0] = next();
    else if (a == "--gy") o.g[1] = next();
    else if (a == "--gz") o.g[2] = next();
    else if (a == "--pr") o.pr = next();
    else if (a == "--pc") o.pc = next();
    else if (a == "--backend") o.backend = next();
    else if (a == "--warmup") o.warmup = next();
    else if (a == "--trials") o.trials = next();
    else if (a == "--double") o.dbl = true;
    else if (a == "--default-layout") o.contiguous = false;
    else if (a == "-o") o.out_of_place = true;
    else if (a == "--no-spectrum-check") o.spectrum = false;
    else {
      fprintf(stderr, "unknown option %s\n", a.c_str());
      return 2;
    }
  }
  int rank = 0, nranks = 1;
  for (const char* v : {"RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK"})
    if (getenv(v)) {
      rank = atoi(getenv(v));
      break;
    }
  for (const char* v : {"WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE"})
    if (getenv(v)) {
      nranks = atoi(getenv(v));
      break;
    }
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  CHECK_HIP(hipSetDevice(rank % ndev));
  return o.dbl ? run<double>(o, rank, nranks) : run<float>(o, rank, nranks);
}
