// halo_test.cpp -- UpdateHalos{X,Y,Z} for dims 0, 1, 2 in sequence through the public C API, whole-pencil compare with
// the closed form (halos, edges, corners, padding); command line, test-file mode and output protocol of the reference's
// tests/cc/halo_test.cc (see native_test.h).
//
//   --gx --gy --gz N      global grid          --pr --pc N   process grid (0 0 = autotune on halos)
//   --rank-order 0|1|2    --backend B  halo backend enum (0 = autotune)      --ac 0|1  axis-contiguous pencils
//   --gd a b c            gdims_dist = g - (a b c)
//   --hex --hey --hez N   halo extent per dimension        --hpx --hpy --hpz 0|1  periodicity per dimension
//   --pdx --pdy --pdz N   padding per dimension            --ax 0|1|2  pencil axis
//   --mem_order a b c     memory order of the tested pencil       -m  accepted, ignored       -f|--testfile FILE
#include "native_test.h"

static int runCase(cudecompHandle_t handle, const Options& o, bool silent) {
  const int rank = worldRank();
  const std::array<int, 3> g = {o.geti("gx", 256), o.geti("gy", 256), o.geti("gz", 256)};
  const std::array<int, 3> gd = o.get3("gd", {0, 0, 0});
  const std::array<int, 3> halo = {o.geti("hex", 1), o.geti("hey", 1), o.geti("hez", 1)};
  const std::array<bool, 3> periods = {o.geti("hpx", 1) != 0, o.geti("hpy", 1) != 0, o.geti("hpz", 1) != 0};
  const std::array<int, 3> pad = {o.geti("pdx", 0), o.geti("pdy", 0), o.geti("pdz", 0)};
  const int axis = o.geti("ax", 0), backend = o.geti("backend", 0);
  if (axis < 0 || axis > 2) throw TestFailure("--ax out of range");

  cudecompGridDescConfig_t config;
  T_CHECK_CD(cudecompGridDescConfigSetDefaults(&config));
  config.pdims[0] = o.geti("pr", 0);
  config.pdims[1] = o.geti("pc", 0);
  config.rank_order = (cudecompRankOrder_t)o.geti("rank-order", 0);
  for (int i = 0; i < 3; ++i) {
    config.gdims[i] = g[i];
    config.gdims_dist[i] = g[i] - gd[i];
    config.transpose_axis_contiguous[i] = o.geti("ac", 0) != 0;
  }
  if (o.has("mem_order")) {
    // the tested pencil gets the requested order, the other two a valid default
    for (int ax = 0; ax < 3; ++ax)
      for (int i = 0; i < 3; ++i) config.transpose_mem_order[ax][i] = (ax == axis) ? o.geti("mem_order", i, i) : i;
  }
  cudecompGridDescAutotuneOptions_t options;
  T_CHECK_CD(cudecompGridDescAutotuneOptionsSetDefaults(&options));
  options.dtype = kDtype;
  options.grid_mode = CUDECOMP_AUTOTUNE_GRID_HALO;
  options.halo_axis = axis;
  for (int i = 0; i < 3; ++i) {
    options.halo_extents[i] = halo[i];
    options.halo_periods[i] = periods[i];
    options.halo_padding[i] = pad[i];
  }
  if (backend != 0) config.halo_comm_backend = (cudecompHaloCommBackend_t)backend;
  else options.autotune_halo_backend = true;

  cudecompGridDesc_t gdesc;
  T_CHECK_CD(cudecompGridDescCreate(handle, &gdesc, &config, &options));
  if (!silent && rank == 0)
    printf("running on %d x %d x %d spatial grid, %d x %d process grid, %s halo backend...\n", g[0], g[1], g[2],
           config.pdims[0], config.pdims[1], cudecompHaloCommBackendToString(config.halo_comm_backend));

  int failures = 0;
  elem_t *data = nullptr, *work = nullptr;
  try {
    cudecompPencilInfo_t p;
    T_CHECK_CD(cudecompGetPencilInfo(handle, gdesc, &p, axis, halo.data(), pad.data()));
    int64_t ws = 0;
    T_CHECK_CD(cudecompGetHaloWorkspaceSize(handle, gdesc, axis, halo.data(), &ws));
    data = TestBuffer::get(0, p.size);
    T_CHECK_CD(cudecompMalloc(handle, gdesc, (void**)&work, std::max<int64_t>(ws, 1) * sizeof(elem_t)));

    std::vector<elem_t> init, ref, host(p.size);
    fillPencil(init, p, g, false, periods);
    fillPencil(ref, p, g, true, periods);
    uploadPencil(data, init.data(), p.size * sizeof(elem_t));
    // input-integrity gate (native_test.h): what a kernel on the library's stream sees of the upload, BEFORE the updates
    if (rank == worldSize() - 1 && std::getenv("CUDECOMP_TEST_INJECT_STALE_INPUT"))  // self-check of the gate
      T_CHECK_HIP(hipMemset(data + p.size / 2, 0xEE, std::min<int64_t>(100, p.size / 2) * sizeof(elem_t)));
    const bool stale_input = InputGate::get().checkInput("halo", data, init, p.size, 0);
    if (stale_input) ++failures;  // a case whose input never arrived proves nothing about the library: it fails, loudly (DIAG line)
    bool pb[3] = {periods[0], periods[1], periods[2]};
    for (int dim = 0; dim < 3; ++dim) {
      if (axis == 0) T_CHECK_CD(cudecompUpdateHalosX(handle, gdesc, data, work, kDtype, halo.data(), pb, dim, pad.data(), 0));
      else if (axis == 1) T_CHECK_CD(cudecompUpdateHalosY(handle, gdesc, data, work, kDtype, halo.data(), pb, dim, pad.data(), 0));
      else T_CHECK_CD(cudecompUpdateHalosZ(handle, gdesc, data, work, kDtype, halo.data(), pb, dim, pad.data(), 0));
    }
    T_CHECK_HIP(hipDeviceSynchronize());
    T_CHECK_HIP(hipMemcpy(host.data(), data, p.size * sizeof(elem_t), hipMemcpyDeviceToHost));
    InputGate::get().checkDownload("halo", data, host, p.size, 0);
    // a halo update never writes interior cells: they must still hold what was uploaded
    const int64_t interior_bad = countMismatches(host, init, p, true);
    if (interior_bad) {
      ++InputGate::get().interior_overwritten;
      fprintf(stderr, "DIAG rank %d halo: interior overwritten after call: %lld interior cells differ from the upload\n", rank,
              (long long)interior_bad);
    }
    const int64_t bad = countMismatches(host, ref, p, false);
    if (bad) {
      fprintf(stderr, "rank %d: %lld cells differ after the halo updates%s\n", rank, (long long)bad,
              stale_input ? " (the input gate had TRIPPED for this case: stale upload)" : " (input gate: the upload was intact before the call)");
      ++failures;
      diagnoseMismatch("halo", data, host, ref, &init, p, false);
    }
  } catch (...) {
    if (data && !TestBuffer::reuse()) (void)hipFree(data);
    if (work) (void)cudecompFree(handle, gdesc, work);
    (void)cudecompGridDescDestroy(handle, gdesc);
    throw;
  }
  TestBuffer::put(data);
  T_CHECK_CD(cudecompFree(handle, gdesc, work));
  notePaths(handle, gdesc);
  T_CHECK_CD(cudecompGridDescDestroy(handle, gdesc));
  return failures ? 1 : 0;
}

int main(int argc, char** argv) { return nativeMain(argc, argv, runCase); }
