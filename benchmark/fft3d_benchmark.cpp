// fft3d_benchmark.cpp -- distributed 3-D FFT (complex-to-complex or real-to-complex) on pencil-decomposed data: batched
// FFTs with hipFFT/rocFFT on the local pencil + the four cuDecomp transposes, forward then inverse.
//
// This is the canonical CALLER of the transpose path and the counterpart of the reference's
// benchmark/benchmark.cu (flow :489-611, strided plans for non-contiguous axes :378-411, slab shortcuts :340-373,
// the R2C flavour :238-330, FLOP model and timing protocol :499-505,587-590,658, tolerances :23-27).  BASELINE.json
// config 4.  It is a harness around the library, not part of it.
//
//   ./fft3d_benchmark --gx 256 --gy 256 --gz 256 [--pr P --pc Q] [--backend B] [--double] [--r2c]
//                     [--default-layout] [-o] [--warmup W] [--trials T] [--no-spectrum-check] [--no-slab-opt]
//
// Slab shortcuts (on unless --no-slab-opt), decided from the process grid the run ends up with:
//   1 x 1   ("xyz")  ONE 3-D FFT on the X pencil, no transposes at all;
//   1 x Q   ("xy")   the X pencil holds whole x-y planes: one 2-D FFT per plane instead of x lines, transpose, y lines;
//   P x 1   ("yz")   the Y pencil holds whole y-z planes (y and z its two fastest dims, i.e. axis-contiguous layout): one
//                    2-D FFT there, and neither the Y<->Z transposes nor the z pass run.
// --r2c: the field is real; x lines are transformed real-to-complex into gx/2+1 coefficients (the decomposed grid is
// (gx/2+1) x gy x gz complex, the real field lives in the same X pencil as rows of 2*(gx/2+1) reals), the inverse ends
// complex-to-real.
//
// One process per rank; ranks are discovered by the library from RANK/WORLD_SIZE (or PMI_*/OMPI_*) like in
// examples/c/basic_usage.c.  Every rank prints one JSON line; benchmark/run_fft3d.py launches the ranks
// and reduces them.
//
// Checks (all must pass):
//   1. spectrum: the forward transform of a plane wave exp(2*pi*i*(kx x/X + ky y/Y + kz z/Z)) (its real part with --r2c)
//      is X*Y*Z (half of it with --r2c) at (kx,ky,kz) and 0 elsewhere -- verified on the distributed pencils the
//      forward pass ends in, through their global indices;
//   2. round trip: forward + inverse + 1/N scaling reproduces uniform random input within 5e-4 (single) /
//      1e-10 (double) max-abs, the reference's tolerances.
// "spectrum_checksum": a weighted sum of the forward spectrum of the random field, weights from the GLOBAL indices, so
// that runs with different grids / shortcuts can be compared with each other (run_fft3d.py adds the ranks' parts).
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
  bool dbl = false, contiguous = true, out_of_place = false, spectrum = true, slab_opt = true, r2c = false;
  int warmup = 3, trials = 5;
};

static bool holdsAxes(const cudecompPencilInfo_t& p, int a, int b) {  // the two fastest memory dims are global axes a and b
  return (p.order[0] == a && p.order[1] == b) || (p.order[0] == b && p.order[1] == a);
}

template <typename Real>
int run(const Options& o, int rank, int nranks) {
  using C = typename std::conditional<std::is_same<Real, double>::value, hipfftDoubleComplex, hipfftComplex>::type;
  const cudecompDataType_t dtype = o.dbl ? CUDECOMP_DOUBLE_COMPLEX : CUDECOMP_FLOAT_COMPLEX;
  const double tol = o.dbl ? 1e-10 : 5e-4;
  hipStream_t stream = 0;
  const int gxc = o.r2c ? o.g[0] / 2 + 1 : o.g[0];  // x extent of the decomposed (complex) grid

  cudecompHandle_t handle;
  CHECK_CD(cudecompInit(&handle, MPI_COMM_WORLD));
  cudecompGridDescConfig_t config;
  CHECK_CD(cudecompGridDescConfigSetDefaults(&config));
  for (int i = 0; i < 3; ++i) {
    config.gdims[i] = i == 0 ? gxc : o.g[i];
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
  if (o.r2c && px.order[0] != 0) {
    fprintf(stderr, "--r2c needs x as the fastest dim of the X pencil\n");
    return 2;
  }

  C *data = nullptr, *data2 = nullptr, *work = nullptr;
  CHECK_HIP(hipMalloc((void**)&data, nel * sizeof(C)));
  if (o.out_of_place) CHECK_HIP(hipMalloc((void**)&data2, nel * sizeof(C)));
  CHECK_CD(cudecompMalloc(handle, gd, (void**)&work, ws * sizeof(C)));

  // ---- which passes run (slab shortcuts) and their plans -----------------------------------------------------------
  const int pr = config.pdims[0], pc = config.pdims[1];
  const bool slab_xyz = o.slab_opt && pr == 1 && pc == 1;
  const bool slab_xy = o.slab_opt && !slab_xyz && pr == 1 && holdsAxes(px, 0, 1) && (!o.r2c || px.order[0] == 0);
  const bool slab_yz = o.slab_opt && !slab_xyz && pc == 1 && holdsAxes(py, 1, 2);
  const char* slab = slab_xyz ? "xyz" : (slab_xy && slab_yz ? "xy+yz" : (slab_xy ? "xy" : (slab_yz ? "yz" : "none")));
  BlockFFT bx, byz;  // X stage when it covers more than x lines (or is real-to-complex); y-z planes
  AxisFFT fx, fy, fz;
  const bool x_block = slab_xyz || slab_xy || o.r2c;
  if (x_block) bx.create(px, slab_xyz ? 3 : (slab_xy ? 2 : 1), o.dbl, o.r2c ? o.g[0] : 0, stream);
  else fx.create(px, 0, o.dbl, stream);
  if (slab_yz) byz.create(py, 2, o.dbl, 0, stream);
  else if (!slab_xy && !slab_xyz) fy.create(py, 1, o.dbl, stream);
  if (!slab_yz && !slab_xyz) fz.create(pz, 2, o.dbl, stream);

  C* cur = data;
  auto hop = [&](cudecompResult_t (*fn)(cudecompHandle_t, cudecompGridDesc_t, void*, void*, void*, cudecompDataType_t,
                                         const int32_t*, const int32_t*, const int32_t*, const int32_t*, hipStream_t)) {
    C* dst = o.out_of_place ? (cur == data ? data2 : data) : cur;
    CHECK_CD(fn(handle, gd, cur, dst, work, dtype, nullptr, nullptr, nullptr, nullptr, stream));
    cur = dst;
  };
  auto xStage = [&](int dir) {
    if (x_block) bx.exec(cur, dir);
    else fx.exec(cur, dir, o.dbl);
  };
  auto forward = [&]() {
    xStage(HIPFFT_FORWARD);
    if (slab_xyz) return;
    hop(cudecompTransposeXToY);
    if (slab_yz) byz.exec(cur, HIPFFT_FORWARD);
    else if (!slab_xy) fy.exec(cur, HIPFFT_FORWARD, o.dbl);
    if (slab_yz) return;
    hop(cudecompTransposeYToZ);
    fz.exec(cur, HIPFFT_FORWARD, o.dbl);
  };
  auto inverse = [&]() {
    if (!slab_xyz) {
      if (!slab_yz) {
        fz.exec(cur, HIPFFT_BACKWARD, o.dbl);
        hop(cudecompTransposeZToY);
      }
      if (slab_yz) byz.exec(cur, HIPFFT_BACKWARD);
      else if (!slab_xy) fy.exec(cur, HIPFFT_BACKWARD, o.dbl);
      hop(cudecompTransposeYToX);
    }
    xStage(HIPFFT_BACKWARD);
  };
  const cudecompPencilInfo_t& pend = slab_xyz ? px : (slab_yz ? py : pz);  // where the forward pass ends

  const double N = (double)o.g[0] * o.g[1] * o.g[2];
  const int64_t row = px.shape[0];                       // complex elements per x row of the X pencil
  const int64_t nrows = px.size / row;
  const int64_t nreal_row = o.r2c ? 2 * row : 0;         // reals per row (padded) with --r2c
  std::vector<std::complex<Real>> host(nel);
  Real* host_r = reinterpret_cast<Real*>(host.data());

  // ---- check 1: spectrum of a plane wave ---------------------------------------------------------------------------
  double spec_err = 0;
  const int k[3] = {3 % o.g[0], 5 % o.g[1], 7 % o.g[2]};
  const bool spectrum = o.spectrum && (!o.r2c || (k[0] > 0 && 2 * k[0] < o.g[0]));
  if (spectrum) {
    for (int64_t r = 0; r < nrows; ++r) {
      const int64_t l1 = r % px.shape[1], l2 = r / px.shape[1];
      double base = 2.0 * M_PI * k[px.order[1]] * (double)(l1 + px.lo[1]) / o.g[px.order[1]] +
                    2.0 * M_PI * k[px.order[2]] * (double)(l2 + px.lo[2]) / o.g[px.order[2]];
      if (o.r2c) {
        for (int64_t x = 0; x < nreal_row; ++x)
          host_r[r * nreal_row + x] = x < o.g[0] ? (Real)std::cos(base + 2.0 * M_PI * k[0] * (double)x / o.g[0]) : (Real)0;
      } else {
        for (int64_t x = 0; x < row; ++x) {
          const double phase = base + 2.0 * M_PI * k[px.order[0]] * (double)(x + px.lo[0]) / o.g[px.order[0]];
          host[r * row + x] = std::complex<Real>((Real)std::cos(phase), (Real)std::sin(phase));
        }
      }
    }
    cur = data;
    CHECK_HIP(hipMemcpy(cur, host.data(), px.size * sizeof(C), hipMemcpyHostToDevice));
    forward();
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(host.data(), cur, pend.size * sizeof(C), hipMemcpyDeviceToHost));
    const double peak_value = o.r2c ? N / 2 : N;
    for (int64_t i = 0; i < pend.size; ++i) {
      int64_t l[3] = {i % pend.shape[0], i / pend.shape[0] % pend.shape[1], i / ((int64_t)pend.shape[0] * pend.shape[1])};
      bool peak = true;
      for (int m = 0; m < 3; ++m) peak = peak && (l[m] + pend.lo[m] == k[pend.order[m]]);
      const std::complex<double> expect(peak ? peak_value : 0.0, 0.0);
      spec_err = std::max(spec_err, std::abs(std::complex<double>(host[i]) - expect) / N);
    }
  }

  // ---- check 2 + timing: random data, forward + inverse ------------------------------------------------------------
  std::vector<std::complex<Real>> ref(px.size);
  Real* ref_r = reinterpret_cast<Real*>(ref.data());
  std::default_random_engine rng(1234 + rank);
  std::uniform_real_distribution<Real> dist(0, 1);
  if (o.r2c) {
    for (int64_t r = 0; r < nrows; ++r)
      for (int64_t x = 0; x < nreal_row; ++x) ref_r[r * nreal_row + x] = x < o.g[0] ? dist(rng) : (Real)0;
  } else {
    for (auto& v : ref) v = std::complex<Real>(dist(rng), dist(rng));
  }
  cur = data;
  CHECK_HIP(hipMemcpy(cur, ref.data(), px.size * sizeof(C), hipMemcpyHostToDevice));

  // weighted checksum of the forward spectrum (weights from global indices: comparable across grids and shortcuts);
  // the field is restored afterwards
  std::complex<double> checksum(0, 0);
  double abs_sum = 0;
  {
    forward();
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(host.data(), cur, pend.size * sizeof(C), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < pend.size; ++i) {
      int64_t l[3] = {i % pend.shape[0], i / pend.shape[0] % pend.shape[1], i / ((int64_t)pend.shape[0] * pend.shape[1])};
      int64_t gk[3];
      for (int m = 0; m < 3; ++m) gk[pend.order[m]] = l[m] + pend.lo[m];
      const double w = (double)((gk[0] * 31 + gk[1] * 17 + gk[2] * 7) % 97 + 1) / 97.0;
      checksum += w * std::complex<double>(host[i]);
      abs_sum += w * std::abs(std::complex<double>(host[i]));
    }
    cur = data;
    CHECK_HIP(hipMemcpy(cur, ref.data(), px.size * sizeof(C), hipMemcpyHostToDevice));
  }

  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  std::vector<double> times;
  const long long nscale = o.r2c ? (long long)px.size * 2 : (long long)px.size;
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
    // untimed, as in the reference
    if (o.r2c) scale_real_kernel<<<(unsigned)((nscale + 255) / 256), 256, 0, stream>>>((Real*)cur, (Real)(1.0 / N), nscale);
    else scale_kernel<<<(unsigned)((px.size + 255) / 256), 256, 0, stream>>>(cur, 1.0 / N, (long long)px.size);
  }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(host.data(), cur, px.size * sizeof(C), hipMemcpyDeviceToHost));
  double rt_err = 0;
  if (o.r2c) {
    for (int64_t r = 0; r < nrows; ++r)
      for (int64_t x = 0; x < o.g[0]; ++x)
        rt_err = std::max<double>(rt_err, std::abs((double)host_r[r * nreal_row + x] - (double)ref_r[r * nreal_row + x]));
  } else {
    for (int64_t i = 0; i < px.size; ++i) rt_err = std::max<double>(rt_err, std::abs(host[i] - ref[i]));
  }

  std::sort(times.begin(), times.end());
  double avg = 0;
  for (double t : times) avg += t;
  avg /= times.size();
  const double gflop = 5.0 * N * std::log2(N) * 1e-9;
  const bool ok = rt_err <= tol && (!spectrum || spec_err <= (o.dbl ? 1e-12 : 1e-4));
  printf("{\"rank\": %d, \"nranks\": %d, \"gdims\": [%d, %d, %d], \"pdims\": [%d, %d], \"backend\": \"%s\", "
         "\"dtype\": \"%s\", \"mode\": \"%s\", \"slab\": \"%s\", \"layout\": \"%s\", \"out_of_place\": %s, \"ms_min\": %.4f, "
         "\"ms_avg\": %.4f, \"ms_max\": %.4f, \"gflops\": %.1f, \"roundtrip_max_abs_err\": %.3e, \"spectrum_rel_err\": %.3e, "
         "\"spectrum_checked\": %s, \"spectrum_checksum\": [%.10e, %.10e], \"spectrum_abs_sum\": %.10e, "
         "\"tolerance\": %.1e, \"ok\": %s}\n",
         rank, nranks, o.g[0], o.g[1], o.g[2], config.pdims[0], config.pdims[1],
         cudecompTransposeCommBackendToString(config.transpose_comm_backend), o.dbl ? "c128" : "c64", o.r2c ? "r2c" : "c2c", slab,
         o.contiguous ? "axis-contiguous" : "default", o.out_of_place ? "true" : "false", times.front(), avg,
         times.back(), gflop / (avg * 1e-3), rt_err, spec_err, spectrum ? "true" : "false", checksum.real(), checksum.imag(),
         abs_sum, tol, ok ? "true" : "false");
  fflush(stdout);

  bx.destroy();
  byz.destroy();
  for (AxisFFT* f : {&fx, &fy, &fz})
    if (f->plan) hipfftDestroy(f->plan);
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
    else if (a == "--no-slab-opt") o.slab_opt = false;
    else if (a == "--r2c") o.r2c = true;
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
