// transpose_test.cpp -- X->Y->Z->Y->X through the public C API with a check after every hop; command line, test-file
// mode and output protocol of the reference's tests/cc/transpose_test.cc (see native_test.h).
//
//   --gx --gy --gz N         global grid (default 256 each)
//   --pr --pc N              process grid (0 0 = autotune)
//   --rank-order 0|1|2       default / row-major / column-major
//   --backend B              transpose backend enum (0 = autotune)
//   --acx --acy --acz 0|1    axis-contiguous pencils
//   --gd a b c               gdims_dist = g - (a b c)
//   --hex|--hey|--hez a b c  halo extents of the X / Y / Z pencils     --pdx|--pdy|--pdz a b c  padding
//   --mem_order 9 ints       transpose_mem_order
//   -o                       out of place
//   -m                       data buffers from cudecompMalloc instead of hipMalloc (the reference's -m selects managed
//                            memory, which has no counterpart here; with NVSHMEM_SM such buffers take the direct put)
//   -f|--testfile FILE       one case per line
// Environment (diagnostics of the shared-GPU hunt, see native_test.h): CUDECOMP_TEST_SENTINEL=1 pre-fills every out-of-place
// output with 0xEE bytes; CUDECOMP_TEST_REUSE_BUFFERS=1 keeps the data buffers for the whole process; a failing hop always
// prints two DIAG lines (what the wrong cells hold, who can see the right ones).  The input-integrity gate (native_test.h,
// on by default) checks the uploaded pencil with a kernel on the library's stream before the first hop and every download
// against the device's view after it.
#include "native_test.h"

static int runCase(cudecompHandle_t handle, const Options& o, bool silent) {
  const int rank = worldRank();
  const std::array<int, 3> g = {o.geti("gx", 256), o.geti("gy", 256), o.geti("gz", 256)};
  const std::array<int, 3> z3 = {0, 0, 0};
  const std::array<int, 3> gd = o.get3("gd", z3);
  const std::array<int, 3> halo[3] = {o.get3("hex", z3), o.get3("hey", z3), o.get3("hez", z3)};
  const std::array<int, 3> pad[3] = {o.get3("pdx", z3), o.get3("pdy", z3), o.get3("pdz", z3)};
  const bool oop = o.has("o") || o.has("out-of-place");
  const bool library_data = o.has("m");
  const int backend = o.geti("backend", 0);

  cudecompGridDescConfig_t config;
  T_CHECK_CD(cudecompGridDescConfigSetDefaults(&config));
  config.pdims[0] = o.geti("pr", 0);
  config.pdims[1] = o.geti("pc", 0);
  config.rank_order = (cudecompRankOrder_t)o.geti("rank-order", 0);
  for (int i = 0; i < 3; ++i) {
    config.gdims[i] = g[i];
    config.gdims_dist[i] = g[i] - gd[i];
  }
  config.transpose_axis_contiguous[0] = o.geti("acx", 0) != 0;
  config.transpose_axis_contiguous[1] = o.geti("acy", 0) != 0;
  config.transpose_axis_contiguous[2] = o.geti("acz", 0) != 0;
  if (o.has("mem_order"))
    for (int i = 0; i < 9; ++i) config.transpose_mem_order[i / 3][i % 3] = o.geti("mem_order", -1, i);
  cudecompGridDescAutotuneOptions_t options;
  T_CHECK_CD(cudecompGridDescAutotuneOptionsSetDefaults(&options));
  options.dtype = kDtype;
  for (int i = 0; i < 4; ++i) options.transpose_use_inplace_buffers[i] = !oop;
  if (backend != 0) config.transpose_comm_backend = (cudecompTransposeCommBackend_t)backend;
  else options.autotune_transpose_backend = true;

  cudecompGridDesc_t gdesc;
  phaseTimes().start();
  T_CHECK_CD(cudecompGridDescCreate(handle, &gdesc, &config, &options));
  phaseTimes().mark(0);
  if (!silent && rank == 0)
    printf("running on %d x %d x %d spatial grid, %d x %d process grid, %s transpose backend...\n", g[0], g[1], g[2],
           config.pdims[0], config.pdims[1], cudecompTransposeCommBackendToString(config.transpose_comm_backend));

  int failures = 0;
  elem_t *data = nullptr, *data2 = nullptr, *work = nullptr;
  try {
    cudecompPencilInfo_t p[3];
    for (int ax = 0; ax < 3; ++ax) T_CHECK_CD(cudecompGetPencilInfo(handle, gdesc, &p[ax], ax, halo[ax].data(), pad[ax].data()));
    int64_t ws = 0;
    T_CHECK_CD(cudecompGetTransposeWorkspaceSize(handle, gdesc, &ws));
    const int64_t nel = std::max(std::max(p[0].size, p[1].size), p[2].size);
    if (library_data) {
      T_CHECK_CD(cudecompMalloc(handle, gdesc, (void**)&data, nel * sizeof(elem_t)));
      if (oop) T_CHECK_CD(cudecompMalloc(handle, gdesc, (void**)&data2, nel * sizeof(elem_t)));
    } else {
      data = TestBuffer::get(0, nel);
      if (oop) data2 = TestBuffer::get(1, nel);
    }
    T_CHECK_CD(cudecompMalloc(handle, gdesc, (void**)&work, std::max<int64_t>(ws, 1) * sizeof(elem_t)));
    phaseTimes().mark(1);

    const std::array<bool, 3> none = {false, false, false};
    std::vector<elem_t> ref[3], host(nel);
    for (int ax = 0; ax < 3; ++ax) fillPencil(ref[ax], p[ax], g, false, none);
    uploadPencil(data, ref[0].data(), p[0].size * sizeof(elem_t));
    // input-integrity gate (native_test.h): what a kernel on the library's stream sees of the upload, BEFORE the first hop
    if (rank == worldSize() - 1 && std::getenv("CUDECOMP_TEST_INJECT_STALE_INPUT"))  // self-check of the gate
      T_CHECK_HIP(hipMemset(data + p[0].size / 2, 0xEE, std::min<int64_t>(100, p[0].size / 2) * sizeof(elem_t)));
    const bool stale_input = InputGate::get().checkInput("XToY", data, ref[0], p[0].size, 0);
    if (stale_input) ++failures;  // a case whose input never arrived proves nothing about the library: it fails, loudly (DIAG line)
    phaseTimes().mark(2);

    struct Hop {
      const char* name;
      cudecompResult_t (*fn)(cudecompHandle_t, cudecompGridDesc_t, void*, void*, void*, cudecompDataType_t, const int32_t*,
                             const int32_t*, const int32_t*, const int32_t*, hipStream_t);
      int from, to;
    };
    const Hop hops[4] = {{"XToY", cudecompTransposeXToY, 0, 1}, {"YToZ", cudecompTransposeYToZ, 1, 2},
                         {"ZToY", cudecompTransposeZToY, 2, 1}, {"YToX", cudecompTransposeYToX, 1, 0}};
    elem_t *in = data, *out = oop ? data2 : data;
    const bool sentinel = oop && sentinelRequested();
    int hop_index = 0;
    for (const Hop& h : hops) {
      if (sentinel) T_CHECK_HIP(hipMemset(out, 0xEE, p[h.to].size * sizeof(elem_t)));
      T_CHECK_CD(h.fn(handle, gdesc, in, out, work, kDtype, halo[h.from].data(), halo[h.to].data(), pad[h.from].data(),
                      pad[h.to].data(), 0));
      phaseTimes().mark(3);
      T_CHECK_HIP(hipDeviceSynchronize());
      phaseTimes().mark(4);
      if (hop_index == 0 && rank == worldSize() - 1 && std::getenv("CUDECOMP_TEST_INJECT_FAULT"))  // exercises the DIAG path
        T_CHECK_HIP(hipMemset(out + p[h.to].size / 2, 0xEE, std::min<int64_t>(1000, p[h.to].size / 2) * sizeof(elem_t)));
      host.resize(p[h.to].size);
      T_CHECK_HIP(hipMemcpy(host.data(), out, p[h.to].size * sizeof(elem_t), hipMemcpyDeviceToHost));
      InputGate::get().checkDownload(h.name, out, host, p[h.to].size, 0);
      const int64_t bad = countMismatches(host, ref[h.to], p[h.to], true);
      if (bad) {
        fprintf(stderr, "rank %d: %s: %lld interior cells differ%s\n", rank, h.name, (long long)bad,
                stale_input ? " (the input gate had TRIPPED for this case: stale upload)" : " (input gate: the upload was intact before the call)");
        ++failures;
        // what `out` held before this hop (out of place, no sentinel): the pencil written two hops earlier
        const std::vector<elem_t>* previous = (oop && !sentinel && hop_index >= 1) ? &ref[hops[hop_index - 1].from] : nullptr;
        diagnoseMismatch(h.name, out, host, ref[h.to], previous, p[h.to], true);
        // Is the INPUT pencil intact now?  (Out of place only: it is still there.)  Hypothesis to check with the next
        // failure: the synchronous upload of the test program had not fully landed when the hop read it.
        if (oop) {
          std::vector<elem_t> again(p[h.from].size);
          T_CHECK_HIP(hipMemcpy(again.data(), in, p[h.from].size * sizeof(elem_t), hipMemcpyDeviceToHost));
          const int64_t in_bad = countMismatches(again, ref[h.from], p[h.from], true);
          fprintf(stderr, "DIAG rank %d %s: input pencil now: %lld wrong interior cells\n", rank, h.name, (long long)in_bad);
        }
      }
      ++hop_index;
      phaseTimes().mark(5);
      if (oop) std::swap(in, out);
    }
  } catch (...) {
    if (library_data) {
      if (data) (void)cudecompFree(handle, gdesc, data);
      if (data2) (void)cudecompFree(handle, gdesc, data2);
      data = data2 = nullptr;
    }
    if (data && !TestBuffer::reuse()) (void)hipFree(data);
    if (data2 && !TestBuffer::reuse()) (void)hipFree(data2);
    if (work) (void)cudecompFree(handle, gdesc, work);
    (void)cudecompGridDescDestroy(handle, gdesc);
    throw;
  }
  if (library_data) {
    T_CHECK_CD(cudecompFree(handle, gdesc, data));
    if (data2) T_CHECK_CD(cudecompFree(handle, gdesc, data2));
  } else {
    TestBuffer::put(data);
    TestBuffer::put(data2);
  }
  T_CHECK_CD(cudecompFree(handle, gdesc, work));
  notePaths(handle, gdesc);
  phaseTimes().mark(6);
  T_CHECK_CD(cudecompGridDescDestroy(handle, gdesc));
  phaseTimes().mark(7);
  return failures ? 1 : 0;
}

int main(int argc, char** argv) { return nativeMain(argc, argv, runCase); }
